"""The cell map and the key-frame analysis held to the REFERENCE'S OWN classes (Points_cloud_cell, Points_cloud_map, Maps_keyframe of
source/cell_map_keyframe.hpp, compiled verbatim into oracle/_ref/libll_ref_cells.so -- oracle/ref_cells.py, recipe `make -C oracle ref`).

  CPU tier   oracle/orc_cellmap.py (and the device arithmetic compiled for the host, tests/hostcheck) against the committed fixtures
             tests/golden/ref_cells*.npz that the reference library wrote (tests/golden/gen_ref_cells.py); where the library itself is
             present (this container), also live against it on fresh seeds, and the fixtures are checked to be what it writes today.
  GPU tier   the cm_* kernels through the C-ABI (api.Cell_map) against the same fixtures.

What is compared exactly: cell sets, per-cell points in insertion order, m_last_update_frame_idx / m_current_frame_idx (including the
first cloud's double increment, CMK:615 + 667), cell_vec of append_cloud (every cell of the first cloud, then cells with >= 3 points),
find_cells_in_radius, the float moments (mean, covariance) of determine_feature, vector counts and non-zero ratios of the images.
What is compared to tolerance: eigenvalues / vectors / labels (the eigen solver is a third-party routine: Eigen's in the reference, a
stand-in in the library, LAPACK in the oracle, a Jacobi sweep on the device -- labels are compared away from their decision thresholds),
the key-frame centre and roi range (the reference sums its cells in the order of their heap addresses: a few float ulps), and the
images up to the sign of the eigen frame's axes (a mirror of either image axis; the eigen solver picks it) at float rounding."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = ["ref_cells0.npz", "ref_cells1.npz"]
F = np.float32


def bits(a):
    return np.ascontiguousarray(a, F).view(np.uint32)


def frames_of(g):
    off = np.concatenate([[0], np.cumsum(g["frame_sizes"])])
    return [g["frames"][off[i]:off[i + 1]] for i in range(len(g["frame_sizes"]))]


def rows(flat, off, i):
    return flat[off[i]:off[i + 1]]


def mirrored(img):
    return [img, img[::-1], img[:, ::-1], img[::-1, ::-1]]


def image_distance(got, want):
    """smallest max-difference over the four sign choices of the eigen frame's second and third axis"""
    return min(float(np.abs(m - want).max()) for m in mirrored(np.asarray(got)))


def ulps(a, b):
    a, b = np.asarray(a, F), np.asarray(b, F)
    return float(np.max(np.abs(a.astype(np.float64) - b) / np.maximum(np.spacing(np.maximum(np.abs(a), np.abs(b))), 1e-45)))


def centre_close(kf, ref, tag=None):
    """get_center adds the cell centres up in float, in the order of the cells' heap addresses in the reference (a std::set of
    shared_ptr) and in cell order everywhere else: the sums differ by rounding that grows with the cell count and the distance from
    the origin; the roi range inherits it"""
    want_c = np.asarray(ref[f"kf_{tag}_centre"] if tag else ref["centre"], np.float64)
    want_r = float(ref[f"kf_{tag}_roi_range"] if tag else ref["roi_range"])
    tol = 2e-6 * max(1.0, float(np.abs(want_c).max())) * 4
    return bool(np.abs(np.asarray(kf["centre"], np.float64) - want_c).max() <= tol and abs(float(kf["roi_range"]) - want_r) <= 2 * tol)


def check_keyframe(kf, g, tag, n_cells_exact=True):
    nv, want_nv = np.asarray(kf["n_vectors"]), g[f"kf_{tag}_n_vectors"]
    assert np.array_equal(nv[:2], want_nv[:2])
    assert np.array_equal(bits(np.asarray(kf["ratio_nonzero"])[:2]), bits(g[f"kf_{tag}_ratio_nonzero"][:2]))
    assert centre_close(kf, g, tag)
    for w in (0, 1):
        scale = max(1.0, float(g[f"kf_{tag}_images"][w].max()))
        assert image_distance(kf["images"][w], g[f"kf_{tag}_images"][w]) < 5e-6 * scale, (tag, w)
    if np.array_equal(nv[2:], want_nv[2:]):   # same cells inside the roi range (a cell within an ulp of it may fall either side)
        for w in (2, 3):
            scale = max(1.0, float(g[f"kf_{tag}_images"][w].max()))
            assert image_distance(kf["images"][w], g[f"kf_{tag}_images"][w]) < 5e-6 * scale, (tag, w)
    else:
        assert np.abs(nv[2:] - want_nv[2:]).max() <= 2


def check_against_reference_fixture(name, make_map, similarity, touched_of=None, radius_of=None, eigen_tol=3e-6):
    """make_map(resolution, revisit) -> object with append / dump / features / keyframe_images / frame() in the oracle's vocabulary;
    touched_of(map, cloud) -> cell_vec of the append (or None: plain append); radius_of(map, pt, r) -> cells (or None: skipped)."""
    g = np.load(os.path.join(HERE, "golden", name))
    res, rev = float(g["resolution"]), int(g["revisit"])
    m = make_map(res, rev)
    frames = frames_of(g)
    for i, c in enumerate(frames):
        if touched_of is not None:
            got = np.asarray(touched_of(m, c), np.int64).reshape(-1, 3)
            want = rows(g["touched"], g["touched_off"], i)
            assert sorted(map(tuple, got.tolist())) == sorted(map(tuple, want.tolist())), (name, i)
        else:
            m.append(c)
        assert m.frame() == int(g["frame_idx"][i]), (name, i)
    xyz, ijk, start, last = m.dump()
    assert np.array_equal(ijk, g["cell_ijk"]) and np.array_equal(start, g["cell_start"])
    assert np.array_equal(bits(xyz), bits(g["store_xyz"])) and np.array_equal(last, g["cell_last"])
    if radius_of is not None:
        for i, q in enumerate(g["radius_query"]):
            got = np.asarray(radius_of(m, q[:3], float(q[3])), np.int64).reshape(-1, 3)
            assert sorted(map(tuple, got.tolist())) == sorted(map(tuple, rows(g["radius_cells"], g["radius_off"], i).tolist())), (name, i)
        assert len(rows(g["radius_cells"], g["radius_off"], len(g["radius_query"]) - 1)) == len(ijk)   # the 40 m query takes every cell
    # determine_feature( 1 ): moments exactly, the eigen stage to tolerance
    f = m.features()
    cnt = np.diff(start)
    big = cnt >= 5
    assert np.array_equal(bits(f["mean"][big]), bits(g["feat_mean"][big]))
    assert np.array_equal(bits(f["cov"][big]), bits(g["feat_cov"][big]))
    scale = np.abs(g["feat_eval"]).max(1) + 1e-30
    assert (np.abs(f["eigen_val"][big] - g["feat_eval"][big]).max(1) / scale[big]).max() < eigen_tol
    ev = g["feat_eval"].astype(np.float64)
    ctr = (ijk.astype(F) * F(np.float64(F(res)) * 0.5) + F(np.float64(F(res)) * 0.25)).astype(F)   # CMK:559-568, 675-677
    dist = np.linalg.norm(ctr.astype(np.float64) - g["feat_mean"], axis=1)
    lim = float(F(np.float64(F(res)) * 0.5)) * 0.75
    margin = np.minimum.reduce([np.abs(dist - lim) / lim, np.abs(ev[:, 1] / 3 - ev[:, 0]) / np.maximum(np.abs(ev[:, 1]), 1e-30),
                                np.abs(ev[:, 2] / 3 - ev[:, 1]) / np.maximum(np.abs(ev[:, 2]), 1e-30)])
    solid = big & (margin > 1e-3)
    assert solid.sum() > 0.8 * big.sum() and np.array_equal(f["type"][solid], g["feat_type"][solid])
    assert np.all(f["type"][~big] == 0) and np.all(g["feat_type"][~big] == 0)
    lab = solid & (g["feat_type"] > 0)
    assert lab.sum() > 200 and np.abs(np.sum(f["vector"][lab] * g["feat_vector"][lab], 1)).min() > 1 - 1e-5
    # Maps_keyframe::analyze
    kf_all = m.keyframe_images(0.9)
    check_keyframe(kf_all, g, "all")
    m0 = make_map(res, 2**31 - 1)
    m0.append(frames[0])
    kf_first = m0.keyframe_images(0.9)
    check_keyframe(kf_first, g, "first")
    # ... over the cells cell_vec named after the first cloud: a key frame holds cells of the full map (LM:1529-1562, CMK:1243-1261)
    later = set(map(tuple, g["later_cells"].tolist()))
    keep = np.array([tuple(c) in later for c in ijk.tolist()], bool)
    idx = np.concatenate([np.arange(start[c], start[c + 1]) for c in np.flatnonzero(keep)])
    m1 = make_map(res, 2**31 - 1)
    m1.append(np.c_[xyz[idx], np.zeros(len(idx), F)].astype(F))
    check_keyframe(m1.keyframe_images(0.9), g, "later")
    # max_similiarity_of_two_image: on the reference's own images (the measure alone), and end to end
    assert abs(similarity(g["kf_all_images"][1], g["kf_first_images"][1]) - float(g["sim_plane"])) < 2e-6
    assert abs(similarity(g["kf_all_images"][0], g["kf_first_images"][0]) - float(g["sim_line"])) < 2e-6
    for close in (getattr(m, "close", None), getattr(m0, "close", None), getattr(m1, "close", None)):
        if close:
            close()


class OracleCells:
    def __init__(self, resolution, revisit):
        from oracle.orc_cellmap import CellMap
        self.m = CellMap(resolution, revisit)
        for name in ("append", "dump", "features", "keyframe_images", "cells_in_radius"):
            setattr(self, name, getattr(self.m, name))

    def frame(self):
        return self.m.frame


# ------------------------------------------------------------------------------------------------------ CPU tier
@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_cell_map_reproduces_the_reference_fixture(name):
    from oracle.orc_cellmap import CellMap
    check_against_reference_fixture(name, OracleCells, CellMap.max_similarity, touched_of=lambda m, c: m.append(c),
                                    radius_of=lambda m, p, r: m.cells_in_radius(p, r), eigen_tol=1e-6)


@pytest.mark.parametrize("name", FIXTURES)
def test_device_math_on_host_reproduces_the_reference_fixture(name):
    from oracle.orc_cellmap import CellMap
    from tests.hostcheck import hc

    class Host(hc.CellMap):
        def frame(self):
            return self.sizes()[2]

    check_against_reference_fixture(name, Host, CellMap.max_similarity)


def test_fixtures_are_what_the_reference_library_writes_today():
    """(only where /root/reference exists) the committed fixtures are the library's current answers, not a stale copy"""
    from oracle import ref_cells as rc
    if not rc.available():
        pytest.skip("oracle/_ref/libll_ref_cells.so not built and no /root/reference here")
    for name in FIXTURES:
        g = np.load(os.path.join(HERE, "golden", name))
        m = rc.RefCellMap(float(g["resolution"]), int(g["revisit"]))
        for i, c in enumerate(frames_of(g)):
            t = m.append(c)
            assert np.array_equal(t, rows(g["touched"], g["touched_off"], i)) and m.frame_idx() == int(g["frame_idx"][i])
        ijk, cnt, last = m.cells()
        assert np.array_equal(ijk, g["cell_ijk"]) and np.array_equal(cnt, np.diff(g["cell_start"])) and np.array_equal(last, g["cell_last"])
        kf = rc.RefKeyframe()
        kf.add_cells(m, ijk)
        a = kf.analyze()
        assert np.array_equal(a["n_vectors"], g["kf_all_n_vectors"]) and image_distance(a["images"][1], g["kf_all_images"][1]) < 1e-6


@pytest.mark.parametrize("seed,resolution,revisit", [(1, 1.0, 3), (2, 0.6, 2**31 - 1), (3, 2.0, 1), (4, 1.0, 2)])
def test_oracle_cell_map_against_the_live_reference(seed, resolution, revisit):
    """fresh seeds, cell by cell, straight against the reference classes (this container only)"""
    from oracle import ref_cells as rc
    from oracle.orc_cellmap import CellMap
    if not rc.available():
        pytest.skip("oracle/_ref/libll_ref_cells.so not built and no /root/reference here")
    rng = np.random.default_rng(seed)
    o, r = CellMap(resolution, revisit), rc.RefCellMap(resolution, revisit)
    base = rng.uniform(-50, 50, 3)
    for k in range(8):
        n = int(rng.integers(0, 1500))
        c = base + rng.normal(0, 1.0 + 0.7 * (k % 3), (n, 3)) * np.array([3.0, 3.0, 0.5])
        c[: n // 2, :2] = base[:2] + rng.uniform(-2.5, 2.5, (n // 2, 2))                          # half of it on a tilted floor patch
        c[: n // 2, 2] = base[2] + 0.1 * (k % 2) + 0.1 * (c[: n // 2, 0] - base[0]) + rng.normal(0, 0.005, n // 2)
        c = c.astype(F)
        if k in (2, 5):
            c = c[: k - 2]                                    # an empty and a three-point cloud
        if k == 6:
            c = (c + np.array([25.0, 0, 0], F)).astype(F)     # somewhere else: revisit counters of the first place run on
        to = np.asarray(o.append(np.c_[c, np.zeros(len(c), F)]), np.int64).reshape(-1, 3)
        tr = r.append(c)
        assert np.array_equal(to, tr) and o.frame == r.frame_idx(), k
    xyz, ijk, start, last = o.dump()
    rijk, rcnt, rlast = r.cells()
    assert np.array_equal(ijk, rijk) and np.array_equal(np.diff(start), rcnt) and np.array_equal(last, rlast)
    f = o.features()
    for c in range(0, len(ijk), 3):
        assert np.array_equal(bits(r.cell_points(ijk[c])), bits(xyz[start[c]:start[c + 1]]))
        rf = r.feature(ijk[c])
        assert np.array_equal(bits(rf["mean"]), bits(f["mean"][c]))
        if rf["n"] >= 5:
            cv = rf["cov"]
            assert np.array_equal(bits([cv[0, 0], cv[0, 1], cv[0, 2], cv[1, 1], cv[1, 2], cv[2, 2]]), bits(f["cov"][c]))
            assert np.abs(rf["eigen_val"] - f["eigen_val"][c]).max() <= 1e-6 * np.abs(rf["eigen_val"]).max()
            if f["margin"][c] > 1e-3:
                assert rf["type"] == f["type"][c]
    for q in range(5):
        p, rad = (base + rng.uniform(-5, 5, 3)).astype(F), float(rng.uniform(0.5, 12.0))
        assert np.array_equal(np.asarray(o.cells_in_radius(p, rad), np.int64).reshape(-1, 3), r.cells_in_radius(p, rad))
    kf = rc.RefKeyframe()
    kf.add_cells(r, rijk)
    a, b = kf.analyze(), o.keyframe_images(0.9)
    assert np.array_equal(a["n_vectors"][:2], b["n_vectors"][:2]) and np.array_equal(bits(a["ratio_nonzero"]), bits(b["ratio_nonzero"][:2]))
    assert a["n_vectors"][1] > 10
    assert centre_close(b, a)
    for w in (0, 1):
        assert image_distance(b["images"][w], a["images"][w]) < 2e-6 * max(1.0, float(a["images"][w].max()))
    assert abs(rc.max_similarity(b["images"][1], b["images"][0]) - CellMap.max_similarity(b["images"][1], b["images"][0])) < 2e-6


# ------------------------------------------------------------------------------------------------------ GPU tier
@pytest.mark.gpu
@pytest.mark.parametrize("name", FIXTURES)
def test_hip_cell_map_reproduces_the_reference_fixture(gpu_lib, name):
    from loam_livox_amd.api import Cell_map, keyframe_similarity

    class Dev:
        def __init__(self, resolution, revisit):
            self.m = Cell_map(max_points=1 << 16, resolution=resolution, minimum_revisit_threshold=revisit)
            self.append, self.dump, self.close = self.m.append_cloud, self.m.dump, self.m.close
            self.features, self.keyframe_images = self.m.features, self.m.keyframe_images

        def frame(self):
            return self.m.stats()[2]

    check_against_reference_fixture(name, Dev, keyframe_similarity, touched_of=lambda d, c: d.m.append_cloud_touched(c, 3))
