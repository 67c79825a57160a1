"""Cell ("cube") match mode, SURVEY 8(f) row 2: Points_cloud_map<float> (cell_map_keyframe.hpp:477-790) and the cell
branch of update_buff_for_matching (laser_mapping.hpp:471-513).
CPU tier: the oracle restatement (oracle/orc_cellmap.py) against hand-checked cases, and against the serial host build of
the device arithmetic and store layout (tests/hostcheck).  GPU tier: the device cell map, the history integration and
the mapping loop in mode 1 against the oracle -- stores and match buffers bit-identical."""
import numpy as np
import pytest

from loam_livox_amd import synth
from oracle import orc
from oracle.orc_cellmap import CellMap
from oracle.orc_mapping import History, LaserMapping

IDENT = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def clouds(seed=0, n_frames=8, n=2000, span=3.0):
    rng = np.random.default_rng(seed)
    out = []
    for f in range(n_frames):
        c = rng.uniform(-span, span, (n, 4)).astype(np.float32)
        if f in (4, 5, 6):
            c[:, 0] += 10.0      # the sensor looks elsewhere for three frames: the first cells go stale
        out.append(c)
    return out


def some_pose(k=0):
    q = synth.quat_from_axis_angle(np.array([0.2, -0.1, 1.0]), np.deg2rad(25.0 + 40.0 * k))
    return np.r_[q, [0.2, 0.1, -0.3]]


def same_store(a, b):
    for x, y, name in zip(a, b, ("xyz", "ijk", "start", "last")):
        assert x.shape == y.shape, name
        assert np.array_equal(bits(x), bits(y)) if x.dtype == np.float32 else np.array_equal(x, y), name


# ------------------------------------------------------------------------------------------------------ CPU tier
def test_cell_geometry_and_rounding():
    m = CellMap(1.0)
    assert m.box == np.float32(0.5) and m.half == np.float32(0.25)   # set_resolution halves (CMK:675-677), then CMK:559-560
    # round((p - 0.25) / 0.5): 0.0 -> -0.5 -> -1 (half away from zero), 0.5 -> 0.5 -> 1, 0.49 -> 0
    k, ok = m.cell_index(np.array([[0.0, 0.5, 0.49], [-0.5, 0.26, 1.0], [np.nan, 0, 0], [1e9, 0, 0]], np.float32))
    assert ok.tolist() == [True, True, False, False]
    assert k[0].tolist() == [-1, 1, 0] and k[1].tolist() == [-2, 0, 2]
    assert np.array_equal(m.centre((1, 0, -1)), np.array([0.75, 0.25, -0.25], np.float32))


def test_append_revisit_rule():
    m = CellMap(1.0, minimum_revisit_threshold=3)
    a = np.array([[0.3, 0.3, 0.3, 9.0]], np.float32)
    b = np.array([[5.3, 0.3, 0.3, 9.0]], np.float32)
    m.append(a); m.append(a)                      # frames 0, 2 (the first cloud advances the counter twice: CMK:615 + 667)
    assert len(m.cells) == 1 and len(m.cell_points((0, 0, 0))) == 2 and m.cells[(0, 0, 0)]["last"] == 2
    m.append(b); m.append(b)                      # frames 3, 4
    m.append(np.r_[a, a])                         # frame 5: 5 - 2 >= 3 -> a fresh cell replaces the old one (CMK:742-754)
    assert len(m.cell_points((0, 0, 0))) == 2 and m.cells[(0, 0, 0)]["last"] == 5 and m.frame == 6
    m.append(np.zeros((0, 4), np.float32))        # an empty cloud still advances m_current_frame_idx (CMK:667)
    assert m.frame == 7


def test_fov_and_radius_selection():
    m = CellMap(1.0)
    pts = np.array([[5.1, 0.1, 0.1], [-5.1, 0.1, 0.1], [5.1, 4.1, 0.1], [5.1, 6.1, 0.1], [30.1, 0.1, 0.1]], np.float32)
    m.append(np.c_[pts, np.zeros(5, np.float32)])
    keys = m.select(IDENT, 20.0, 45.0)
    # behind the sensor: out; atan(4.25/5.25) = 39 deg: in; atan(6.25/5.25) = 50 deg: out; 30 m: out of range
    assert keys == [(10, 0, 0), (10, 8, 0)]
    yaw180 = np.r_[synth.quat_from_axis_angle(np.array([0, 0, 1.0]), np.pi), [0, 0, 0]]
    assert m.select(yaw180, 20.0, 45.0) == [(-11, 0, 0)]
    assert len(m.select(IDENT, 100.0, 45.0)) == 3


def test_query_filter_is_a_voxel_grid_per_cell():
    m = CellMap(2.0)
    rng = np.random.default_rng(3)
    c = rng.uniform(0, 4, (3000, 4)).astype(np.float32)
    m.append(c)
    n_before = m.n_points()
    cat, keys = m.query_filter(np.r_[0, 0, 0, 1, -5.0, 2.0, 2.0], 50.0, 60.0, 0.25, down_sample_replace=0)
    assert m.n_points() == n_before
    k, _ = m.cell_index(c[:, :3])
    off = 0
    for key in keys:
        sel = np.all(k == np.array(key), axis=1)
        want = orc.voxel_grid(np.c_[c[sel, :3], np.zeros(sel.sum(), np.float32)], 0.25)[1]
        assert np.array_equal(bits(cat[off:off + len(want)]), bits(want))
        off += len(want)
    assert off == len(cat) and np.all(cat[:, 3] == 0)
    cat2, _ = m.query_filter(np.r_[0, 0, 0, 1, -5.0, 2.0, 2.0], 50.0, 60.0, 0.25, down_sample_replace=1)
    assert np.array_equal(bits(cat2), bits(cat)) and m.n_points() < n_before          # set_pointcloud (LM:492-495)


@pytest.mark.parametrize("replace", [1, 0])
def test_host_build_of_device_store_matches_oracle(replace):
    from tests.hostcheck import hc
    o, h = CellMap(1.0, 3), hc.CellMap(1.0, 3)
    for f, c in enumerate(clouds()):
        o.append(c); h.append(c)
        same_store(o.dump(), h.dump())
        assert h.sizes() == (len(o.cells), o.n_points(), o.frame)
        if f % 2 == 1:
            pose = some_pose(f)
            ca, keys = o.query_filter(pose, 4.0, 45.0, 0.2, replace)
            cb, nsel = h.query_filter(pose, 4.0, 45.0, 0.2, replace)
            assert nsel == len(keys) > 20 and len(ca) > 100
            assert np.array_equal(bits(ca), bits(cb))
            same_store(o.dump(), h.dump())
    # frame 7 came back to the first region after three frames away: those cells started over
    assert o.n_points() < 8 * 2000 - 3000


def test_host_build_large_coordinates_and_fine_leaf():
    from tests.hostcheck import hc
    rng = np.random.default_rng(9)
    o, h = CellMap(0.6), hc.CellMap(0.6)
    c = (rng.uniform(-2, 2, (4000, 4)) + np.array([4321.0, -987.0, 55.0, 0])).astype(np.float32)
    o.append(c); h.append(c)
    same_store(o.dump(), h.dump())
    pose = np.r_[0, 0, 0, 1, 4315.0, -987.0, 55.0]
    ca, keys = o.query_filter(pose, 30.0, 50.0, 0.05, 1)
    cb, nsel = h.query_filter(pose, 30.0, 50.0, 0.05, 1)
    assert nsel == len(keys) > 50 and np.array_equal(bits(ca), bits(cb))
    same_store(o.dump(), h.dump())
    with pytest.raises(ValueError):
        h.query_filter(pose, 30.0, 50.0, 0.0002, 0)   # more than 1020 leaves across one cell


def test_history_feeds_cell_maps_every_frame():
    rng = np.random.default_rng(4)
    h = History(maximum_history_size=2, line_res=0.2, plane_res=0.5)
    h.enable_cell_map(1.0, 5000)
    for k in range(4):
        c = rng.uniform(-3, 3, (200, 4)).astype(np.float32)
        s = rng.uniform(-3, 3, (800, 4)).astype(np.float32)
        assert h.add(c, s, IDENT, t_step=0.5, angle_step=0.1) == (k < 2)   # history full and no motion -> not pushed ...
    assert h.cells[0].frame == 5 and h.cells[1].frame == 5                # ... but appended to the cell maps (LM:1492-1493; +1: first cloud)
    mc, ms = h.refresh_cells(np.r_[0, 0, 0, 1, -6.0, 0, 0], (100.0, 100.0), 45.0, 1)
    assert 0 < len(mc) <= 800 and 0 < len(ms) <= 3200


def structured_cloud(seed=5, offset=(30.0, -20.0, 5.0)):
    """planes, lines and blobs some tens of metres from the origin (the float second moments cancel there)"""
    rng = np.random.default_rng(seed)
    pts = []
    for i in range(40):
        o = rng.uniform(-8, 8, 3) + np.array(offset)
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        u = np.cross(n, [0, 0, 1.0]); u /= np.linalg.norm(u); v = np.cross(n, u)
        if i % 3 == 0:
            p = o + rng.uniform(-1.5, 1.5, (3000, 1)) * u + rng.uniform(-1.5, 1.5, (3000, 1)) * v + rng.normal(0, 0.01, (3000, 3))
        elif i % 3 == 1:
            p = o + rng.uniform(-2, 2, (800, 1)) * u + rng.normal(0, 0.01, (800, 3))
        else:
            p = o + rng.normal(0, 0.3, (500, 3))
        pts.append(p)
    c = np.concatenate(pts).astype(np.float32)
    return np.c_[c, np.zeros(len(c), np.float32)]


def same_features(fo, fd):
    """oracle (LAPACK eigh) vs device arithmetic (double Jacobi): moments bit-identical, eigenvalues to float rounding,
    labels identical away from the decision boundaries, feature vectors equal up to sign"""
    assert np.array_equal(bits(fo["mean"]), bits(fd["mean"])) and np.array_equal(bits(fo["cov"]), bits(fd["cov"]))
    scale = np.abs(fo["eigen_val"]).max(1) + 1e-30
    assert (np.abs(fo["eigen_val"] - fd["eigen_val"]).max(1) / scale).max() < 3e-7
    solid = fo["margin"] > 1e-3
    assert solid.mean() > 0.95 and np.array_equal(fo["type"][solid], fd["type"][solid])
    m = solid & (fo["type"] > 0)
    assert np.abs(np.sum(fo["vector"][m] * fd["vector"][m], 1)).min() > 1 - 1e-5
    assert np.all(fd["vector"][fd["type"] == 0] == 0)
    v = fd["vector"][fd["type"] > 0]
    lead = np.where(v[:, 0] != 0, v[:, 0], np.where(v[:, 1] != 0, v[:, 1], v[:, 2]))
    assert np.all(lead > 0) and np.allclose(np.linalg.norm(v, axis=1), 1.0, atol=1e-6)


def test_cell_features_known_answers():
    m = CellMap(2.0)                                        # cells of 1 m
    g = np.linspace(0.05, 0.95, 10, dtype=np.float32)
    plane = np.array([[x, y, 0.5] for x in g for y in g], np.float32)                   # z = 0.5 inside cell (0,0,0)
    line = np.array([[x + 3.0, 0.5, 0.5] for x in np.linspace(0.05, 0.95, 30)], np.float32)   # along x in cell (3,0,0)
    blob = np.random.default_rng(1).uniform(0.1, 0.9, (200, 3)).astype(np.float32) + np.float32([0, 3, 0])
    few = np.array([[6.2, 0.2, 0.2], [6.4, 0.4, 0.4]], np.float32)                       # fewer than 5 points
    corner = np.array([[9.02 + 0.01 * i, 0.02, 0.02 + 0.001 * i] for i in range(20)], np.float32)   # mean far from the centre
    m.append(np.concatenate([plane, line, blob, few, corner]))
    f = m.features()
    keys = sorted(m.cells)
    got = {k: (int(f["type"][i]), f["vector"][i]) for i, k in enumerate(keys)}
    assert got[(0, 0, 0)][0] == 2 and abs(abs(got[(0, 0, 0)][1][2]) - 1) < 1e-5        # plane, normal = z
    assert got[(3, 0, 0)][0] == 1 and abs(abs(got[(3, 0, 0)][1][0]) - 1) < 1e-5        # line, direction = x
    assert got[(0, 3, 0)][0] == 0 and got[(6, 0, 0)][0] == 0 and got[(9, 0, 0)][0] == 0
    i = keys.index((0, 0, 0))
    assert np.allclose(f["mean"][i], [0.5, 0.5, 0.5], atol=1e-6) and np.allclose(f["eigen_val"][i][0], 0, atol=1e-7)


def test_host_build_of_cell_statistics_matches_oracle():
    from tests.hostcheck import hc
    c = structured_cloud()
    o, h = CellMap(1.0), hc.CellMap(1.0)
    o.append(c); h.append(c)
    fo = o.features()
    assert np.bincount(fo["type"], minlength=3).min() > 50     # all three labels occur
    same_features(fo, h.features())
    o.query_filter(np.r_[0, 0, 0, 1, 20.0, -20.0, 5.0], 40.0, 60.0, 0.1, 1)     # statistics follow the points the cells hold
    h.query_filter(np.r_[0, 0, 0, 1, 20.0, -20.0, 5.0], 40.0, 60.0, 0.1, 1)
    same_features(o.features(), h.features())


def same_keyframe(ko, kd):
    assert ko["near_bin_edge"] == 0                                      # no vector sits on a histogram bin edge
    assert np.array_equal(ko["n_vectors"], kd["n_vectors"]) and ko["n_vectors"].min() > 20
    assert np.array_equal(bits(ko["ratio_nonzero"]), bits(kd["ratio_nonzero"]))
    assert np.array_equal(bits(ko["centre"]), bits(kd["centre"])) and np.float32(ko["roi_range"]) == np.float32(kd["roi_range"])
    assert np.abs(ko["eigen_R"] - kd["eigen_R"]).max() < 1e-6
    for i in range(4):                                                   # blur: float sums here, double in the oracle
        assert np.abs(ko["images"][i] - kd["images"][i]).max() < 2e-6 * max(1.0, float(ko["images"][i].max()))
        assert abs(float(kd["images"][i].sum()) - ko["n_vectors"][i]) < 1e-2        # the Gaussian is normalised, the padding wraps


def test_keyframe_images_host_build_matches_oracle():
    from tests.hostcheck import hc
    c = structured_cloud()
    o, h = CellMap(1.0), hc.CellMap(1.0)
    o.append(c); h.append(c)
    ko = o.keyframe_images(0.9)
    same_keyframe(ko, h.keyframe_images(0.9))
    assert ko["n_vectors"][2] < ko["n_vectors"][0] and ko["n_vectors"][3] < ko["n_vectors"][1]     # the ROI drops the outer cells
    k0 = h.keyframe_images(0.0)                                          # no ROI requested
    assert np.all(k0["images"][2:] == 0) and np.array_equal(bits(k0["images"][:2]), bits(h.keyframe_images(0.9)["images"][:2]))
    # R: orthonormal, right-handed, first axis = dominant plane-normal direction
    R = ko["eigen_R"][0].astype(np.float64)
    assert np.allclose(R.T @ R, np.eye(3), atol=1e-6) and abs(np.linalg.det(R) - 1) < 1e-6


def test_keyframe_similarity_oracle_properties():
    rng = np.random.default_rng(2)
    a = rng.uniform(0, 1, (60, 60)).astype(np.float32) ** 8            # peaky, like a direction histogram
    assert abs(CellMap.max_similarity(a, a) - 1.0) < 1e-12
    assert abs(CellMap.max_similarity(a, np.roll(a, (7, -20), (0, 1))) - 1.0) < 1e-12      # found again under a circular shift
    b = rng.uniform(0, 1, (60, 60)).astype(np.float32) ** 8
    s_ab = CellMap.max_similarity(a, b)
    assert 0 < s_ab < 0.6 and abs(s_ab - CellMap.max_similarity(b, a)) < 1e-12
    assert CellMap.max_similarity(a, np.zeros((60, 60), np.float32)) == 0.0
    # brute force over the 61 x 61 window positions of cv::matchTemplate on the wrap-padded image
    pb = np.pad(b.astype(np.float64), 30, mode="wrap")
    best = max(float((a * pb[y:y + 60, x:x + 60]).sum()) for y in range(61) for x in range(61))
    assert abs(best / np.sqrt((a.astype(np.float64) ** 2).sum() * (b.astype(np.float64) ** 2).sum()) - s_ab) < 1e-9


def keyframe_pair(seed=3):
    """the structured scene as key frame a; the same scene under a small rigid motion, with noise and 80 % overlap, as b"""
    c = structured_cloud(5)
    rng = np.random.default_rng(seed)
    T = np.r_[synth.quat_from_axis_angle(np.array([0.1, 0.2, 1.0]), np.deg2rad(1.5)), [0.35, -0.2, 0.1]]      # frame b -> frame a
    R = synth.quat_to_mat(T[:4])
    xyz = c[:, :3].astype(np.float64)
    keep = rng.uniform(size=len(xyz)) < 0.8
    b = ((xyz[keep] - T[4:]) @ R) + rng.normal(0, 0.005, (int(keep.sum()), 3))
    a = xyz[rng.uniform(size=len(xyz)) < 0.8]
    z = lambda p: np.c_[p, np.zeros(len(p))].astype(np.float32)
    return z(a), z(b), T


def test_oracle_scene_alignment_recovers_the_motion():
    from oracle.orc_scene_alignment import SceneAlignment, keyframe_clouds
    a, b, T = keyframe_pair()
    ka, kb = CellMap(1.0), CellMap(1.0)
    ka.append(a); kb.append(b)
    line_a, plane_a, centre_a = keyframe_clouds(ka)
    assert len(line_a) > 1000 and len(plane_a) > 10000 and np.all(line_a[:, 3] == 0)
    sa = SceneAlignment()
    thr = sa.find_tranfrom_of_two_mappings(ka, kb)
    dt, dr = synth.pose_error(sa.pose, T)
    assert dt < 0.03 and dr < 0.002 and 0 < thr < 0.2                 # from a 1.2 m centre offset to centimetres
    assert len(sa.reports) == 3 and all(r.accepted for r in sa.reports)               # 8x, 4x and 1x resolution (SA:313-352)
    assert sa.reports[0].n_blocks_last < sa.reports[2].n_blocks_last


@pytest.mark.parametrize("seed,res,max_icp,accepted", [(3, 0.4, 10, 0.2), (4, 0.2, 2, 0.35), (5, 0.4, 3, 0.01)])
def test_oracle_scene_alignment_equals_the_reference_driver(tmp_path, seed, res, max_icp, accepted):
    """Scene_alignment::find_tranfrom_of_two_mappings as the loop detector runs it (init() + the driver, scene_alignment.hpp:233-243,
    269-391) compiled VERBATIM on the reference's own Maps_keyframe / Points_cloud_map / Point_cloud_registration
    (tests/verbatim_build.py build_scene_alignment; stand-in third-party headers) against oracle/orc_scene_alignment.py on the same
    two clouds.  (0.2 m / 2 iterations / 0.35 are the loop detector's values, laser_mapping.hpp:700-706; accepted = 0.01 makes the
    coarse-to-fine loop stop early, SA:350.)  The reference concatenates a key frame's cells in the order of their heap addresses,
    so the voxel centroids differ in their last bits: poses to 1e-5, not to the bit."""
    import subprocess
    from oracle.orc_scene_alignment import SceneAlignment
    from tests import verbatim_build
    exe = verbatim_build.build_scene_alignment()
    if not exe:
        pytest.skip("verbatim scene-alignment harness not built (no /root/reference here and none travelled)")
    a, b, _ = keyframe_pair(seed)

    def by_cell(c):
        # The reference walks a key frame's cells in the order of their heap addresses (a std::set of shared_ptr), i.e. -- with a
        # bump allocator and nothing freed in between -- in the order the cells were created = first touched by the cloud.  Clouds
        # sorted by cell (stable: the order inside a cell stays) make that order the oracle's ascending cell order.
        k, ok = CellMap(1.0).cell_index(c[:, :3])
        assert ok.all()
        return c[np.lexsort((np.arange(len(c)), k[:, 2], k[:, 1], k[:, 0]))]
    a, b = by_cell(a), by_cell(b)
    fa, fb, out = str(tmp_path / "a.bin"), str(tmp_path / "b.bin"), str(tmp_path / "out.txt")
    a[:, :3].astype(np.float32).tofile(fa)
    b[:, :3].astype(np.float32).tofile(fb)
    subprocess.check_call([exe, fa, fb, repr(res), repr(res), str(max_icp), repr(accepted), "100000000", out], timeout=600, stdout=subprocess.DEVNULL)
    l0, l1 = open(out).read().strip().split("\n")
    _, icp_final, thr_ref = l0.split()
    pose_ref = np.array([float(v) for v in l1.split()])
    ka, kb = CellMap(1.0), CellMap(1.0)
    ka.append(a); kb.append(b)
    so = SceneAlignment(res, res, max_icp, accepted, 100000000)   # (no random sub-sampling: the reference seeds it from random_device)
    thr = so.find_tranfrom_of_two_mappings(ka, kb)
    dt, dr = synth.pose_error(so.pose, pose_ref)
    assert dt < 1e-5 and dr < 1e-6 and abs(thr - float(thr_ref)) < 1e-5 * max(1.0, abs(thr))
    rounds = len(so.reports)
    assert int(icp_final) == (2 * max_icp if rounds == 3 else max_icp)   # SA:327: doubled once the finest resolution is reached
    if accepted == 0.01:
        assert rounds < 3                                                 # SA:350-351: stopped after a coarse round


@pytest.mark.parametrize("revisit,replace,fov", [(3, 1, 50.0), (2**31 - 1, 0, 30.0), (2, 1, 80.0)])
def test_oracle_cell_refresh_equals_the_reference_branch(tmp_path, revisit, replace, fov):
    """update_buff_for_matching with m_matching_mode = 1 (laser_mapping.hpp:465-537 as ONE verbatim range, with if_pt_in_fov :309-324 and
    the cell-map appends :1492-1493) on the reference's own Points_cloud_map (tests/verbatim_build.py build_cell_refresh; stand-in
    pcl::VoxelGrid / octree) against History.refresh_cells over five frames: cells in range and field of view, per-cell filter, the
    replace of a cell's points by its filtered cloud, concatenation, final filters.  Same voxels every frame; coordinates to an ulp (the
    reference concatenates the cells in octree order, the oracle in cell order: the float sums inside a final voxel differ in order)."""
    import struct
    import subprocess
    from oracle.orc_mapping import History
    from tests import verbatim_build
    exe = verbatim_build.build_cell_refresh()
    if not exe:
        pytest.skip("verbatim cell-refresh harness not built (no /root/reference here and none travelled)")
    rng = np.random.default_rng(3)
    frames = []
    for k in range(5):
        c = structured_cloud(5 + (k % 3))                      # (the third and fourth frame revisit the first two places)
        corner, surf = c[rng.uniform(size=len(c)) < 0.15].astype(np.float32), c[rng.uniform(size=len(c)) < 0.6].astype(np.float32)
        corner[:, 3] = 0
        surf[:, 3] = 0
        pose = np.r_[synth.quat_from_axis_angle(np.array([0.1, 0.2, 1.0]), np.deg2rad(20.0 * k)), [25.0 + 0.5 * k, -22.0, 4.0]]
        frames.append((pose, corner, surf))
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("i", len(frames)))
        for pose, c, s in frames:
            f.write(np.asarray(pose, np.float64).tobytes() + struct.pack("ii", len(c), len(s)) + c.tobytes() + s.tobytes())
    line_res, plane_res, ranges = 0.1, 0.4, (12.0, 15.0)
    subprocess.check_call([exe, fin, fout, "1.0", str(revisit), repr(line_res), repr(plane_res), repr(ranges[0]), repr(ranges[1]), repr(fov), str(replace)],
                          timeout=600, stdout=subprocess.DEVNULL)
    raw, off = open(fout, "rb").read(), 0
    h = History(100, line_res, plane_res)
    h.enable_cell_map(1.0, revisit)
    n_exact = n_all = 0
    for k, (pose, c, s) in enumerate(frames):
        h.cells[0].append(c)
        h.cells[1].append(s)
        got = h.refresh_cells(pose, ranges, fov, replace)
        for kind, leaf in ((0, line_res), (1, plane_res)):
            n = struct.unpack_from("i", raw, off)[0]
            ref = np.frombuffer(raw, np.float32, 4 * n, off + 4).reshape(n, 4)
            off += 4 + 16 * n
            a = got[kind]
            assert len(a) == n > 200, (k, kind)
            ka, kb = np.floor(a[:, :3].astype(np.float64) / leaf).astype(np.int64), np.floor(ref[:, :3].astype(np.float64) / leaf).astype(np.int64)
            ia, ib = np.lexsort((ka[:, 2], ka[:, 1], ka[:, 0])), np.lexsort((kb[:, 2], kb[:, 1], kb[:, 0]))
            assert np.array_equal(ka[ia], kb[ib]) and np.abs(a[ia, :3] - ref[ib, :3]).max() < 2e-5, (k, kind)
            n_exact += int(np.sum(np.all(bits(a[ia, :3]) == bits(ref[ib, :3]), axis=1)))
            n_all += n
    assert off == len(raw) and n_exact > 0.9 * n_all   # (most voxels lie inside one cell: same points, same order, same bits)


# ------------------------------------------------------------------------------------------------------ GPU tier
@pytest.mark.gpu
def test_device_scene_alignment_matches_oracle(gpu_lib):
    from loam_livox_amd.api import Cell_map
    from loam_livox_amd.scene_alignment import Scene_alignment, keyframe_clouds as dev_clouds
    from oracle.orc_scene_alignment import SceneAlignment, keyframe_clouds
    a, b, T = keyframe_pair()
    ka, kb = CellMap(1.0), CellMap(1.0)
    da, db = Cell_map(max_points=1 << 17, resolution=1.0), Cell_map(max_points=1 << 17, resolution=1.0)
    ka.append(a); kb.append(b); da.append_cloud(a); db.append_cloud(b)
    for o, d in ((ka, da), (kb, db)):                                  # the same cells carry the same labels ...
        assert np.array_equal(o.features()["type"], d.features()["type"])
        for x, y in zip(keyframe_clouds(o), dev_clouds(d)):            # ... so the same clouds and centres go in
            assert np.array_equal(bits(x), bits(y))
    so, sd = SceneAlignment(), Scene_alignment()
    thr_o, thr_d = so.find_tranfrom_of_two_mappings(ka, kb), sd.find_tranfrom_of_two_mappings(da, db)
    dt, dr = synth.pose_error(sd.pose, so.pose)
    assert dt < 1e-7 and dr < 1e-7 and abs(thr_o - thr_d) < 1e-9
    assert [r.n_blocks_last for r in sd.reports] == [r.n_blocks_last for r in so.reports]
    assert [r.icp_iterations for r in sd.reports] == [r.icp_iterations for r in so.reports]
    dt, dr = synth.pose_error(sd.pose, T)
    assert dt < 0.03 and dr < 0.002
    da.close(); db.close()


@pytest.mark.gpu
def test_device_keyframe_images_and_similarity(gpu_lib):
    from loam_livox_amd.api import Cell_map, keyframe_similarity
    ca, cb = structured_cloud(5), structured_cloud(6)
    maps = []
    for c in (ca, cb):
        o, d = CellMap(1.0), Cell_map(max_points=1 << 17, resolution=1.0)
        o.append(c); d.append_cloud(c)
        ko, kd = o.keyframe_images(0.9), d.keyframe_images(0.9)
        same_keyframe(ko, kd)
        k0 = d.keyframe_images(0.0)
        assert np.all(k0["images"][2:] == 0) and np.array_equal(bits(k0["images"][:2]), bits(kd["images"][:2]))
        maps.append((ko, kd))
        d.close()
    (oa, da), (ob, db) = maps
    for i in range(4):
        for x, y in ((da["images"][i], da["images"][i]), (da["images"][i], db["images"][i]), (db["images"][i], np.roll(db["images"][i], (11, 5), (0, 1)))):
            want = CellMap.max_similarity(x, y)
            assert abs(keyframe_similarity(x, y) - want) < 1e-5
    assert abs(keyframe_similarity(da["images"][1], da["images"][1]) - 1.0) < 1e-6
    assert keyframe_similarity(da["images"][1], np.zeros((60, 60), np.float32)) == 0.0
    e = Cell_map(max_points=1000, resolution=1.0)
    assert np.all(e.keyframe_images(0.9)["images"] == 0)
    e.close()


@pytest.mark.gpu
def test_device_cell_statistics_match_oracle(gpu_lib):
    from loam_livox_amd.api import Cell_map
    c = structured_cloud()
    o, d = CellMap(1.0), Cell_map(max_points=1 << 17, resolution=1.0)
    o.append(c); d.append_cloud(c)
    same_features(o.features(), d.features())
    pose = np.r_[0, 0, 0, 1, 20.0, -20.0, 5.0]
    o.query_filter(pose, 40.0, 60.0, 0.1, 1); d.query_filter(pose, 40.0, 60.0, 0.1, 1)
    same_features(o.features(), d.features())
    e = Cell_map(max_points=1000, resolution=1.0)
    assert len(e.features()["type"]) == 0
    d.close(); e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("replace", [1, 0])
def test_device_cell_map_bit_exact(gpu_lib, replace):
    from loam_livox_amd.api import Cell_map
    o, d = CellMap(1.0, 3), Cell_map(max_points=40000, resolution=1.0, minimum_revisit_threshold=3)
    for f, c in enumerate(clouds()):
        if f == 2:
            c = c.copy(); c[5, 1] = np.nan; c[6, 2] = np.inf; c[7, 0] = 3.0e6   # dropped: non-finite / beyond the key range
        o.append(c); d.append_cloud(c)
        assert d.stats() == (len(o.cells), o.n_points(), o.frame)
        same_store(o.dump(), d.dump())
        if f % 2 == 1:
            pose = some_pose(f)
            ca, keys = o.query_filter(pose, 4.0, 45.0, 0.2, replace)
            cb, nsel = d.query_filter(pose, 4.0, 45.0, 0.2, replace)
            assert nsel == len(keys) > 20
            assert np.array_equal(bits(ca), bits(cb))
            same_store(o.dump(), d.dump())
    d.append_cloud(np.zeros((0, 4), np.float32)); o.append(np.zeros((0, 4), np.float32))
    assert d.stats()[2] == o.frame
    with pytest.raises(RuntimeError):
        d.query_filter(IDENT, 4.0, 45.0, 0.0002, 0)
    with pytest.raises(RuntimeError):
        d.append_cloud(np.zeros((50000, 4), np.float32))      # exceeds max_points
    d.close()


@pytest.mark.gpu
def test_device_cell_map_large(gpu_lib):
    """1.2 M points in ~10^5 cells: size-independent properties (every point lands in its own cell, order, counts)."""
    from loam_livox_amd.api import Cell_map
    rng = np.random.default_rng(11)
    d = Cell_map(max_points=1 << 21, resolution=1.0)
    o = CellMap(1.0)
    total = 0
    for f in range(3):
        c = rng.uniform(-12, 12, (400000, 4)).astype(np.float32)
        d.append_cloud(c)
        total += len(c)
    nc, npts, fr = d.stats()
    assert npts == total and fr == 4                                            # (the first cloud counts twice, CMK:615 + 667)
    xyz, ijk, start, last = d.dump()
    k, ok = o.cell_index(xyz)
    assert ok.all()
    cell_of_point = np.repeat(np.arange(nc), np.diff(start))
    assert np.array_equal(k, ijk[cell_of_point].astype(np.int64))             # grouped by cell
    key = ((ijk[:, 0].astype(np.int64) + (1 << 20)) << 42) + ((ijk[:, 1].astype(np.int64) + (1 << 20)) << 21) + ijk[:, 2] + (1 << 20)
    assert np.all(np.diff(key) > 0) and last.max() == 3 and np.mean(last == 3) > 0.9   # cells ascending, most touched by the last cloud
    o.cells = {tuple(k3): {"pts": [], "last": 0} for k3 in ijk.tolist()}
    selected = set(o.select(IDENT, 8.0, 40.0))
    n_selected_points = int(sum(start[i + 1] - start[i] for i, k3 in enumerate(map(tuple, ijk.tolist())) if k3 in selected))
    cat, nsel = d.query_filter(IDENT, 8.0, 40.0, 0.25, 1)
    nc2, npts2, _ = d.stats()
    assert nsel == len(selected) and 0 < nsel < nc and nc2 == nc
    assert npts2 == npts - n_selected_points + len(cat) and len(cat) < n_selected_points   # the leaves replaced the points
    assert np.all(cat[:, 3] == 0)
    kc, _ = o.cell_index(cat[:, :3])
    assert all(tuple(r) in selected for r in kc.tolist())                     # a centroid stays inside its cell
    cat2, nsel2 = d.query_filter(IDENT, 8.0, 40.0, 0.25, 1)                   # idempotent: already one point per leaf
    assert nsel2 == nsel and np.array_equal(bits(cat2), bits(cat)) and d.stats()[1] == npts2
    d.close()


@pytest.mark.gpu
def test_device_history_cell_mode_bit_exact(gpu_lib):
    from loam_livox_amd.api import History_buffer, Map_buffer
    rng = np.random.default_rng(4)
    dev, ora = History_buffer(3, 4000, 0.2, 0.5), History(3, 0.2, 0.5)
    dev.enable_cell_map(1 << 16, 1.0, 4)
    ora.enable_cell_map(1.0, 4)
    m = Map_buffer()
    pose = IDENT.copy()
    for k in range(7):
        c = rng.uniform(-6, 6, (300 + 50 * k, 4)).astype(np.float32)
        s = rng.uniform(-6, 6, (3000 + 100 * k, 4)).astype(np.float32)
        if k == 2:
            c = np.zeros((0, 4), np.float32)
        if k in (4, 5):
            pose = synth.pose_compose(pose, np.r_[synth.quat_from_axis_angle(np.array([0, 0, 1.0]), np.deg2rad(1.0 if k == 4 else 8.0)), [0.1, 0, 0]])
        assert dev.add(c, s, pose, 0.5, 0.1) == ora.add(c, s, pose, 0.5, 0.1)
        for kind in range(2):
            same_store(ora.cells[kind].dump(), dev.cell_map(kind).dump())
        view = synth.pose_compose(pose, np.r_[0, 0, 0, 1, -9.0, 0, 0])     # stand back so that the cloud is in the field of view
        nc, ns = dev.refresh_cells(m, view, 12.0, 14.0, 50.0, 1)
        mc, ms = ora.refresh_cells(view, (12.0, 14.0), 50.0, 1)
        assert (nc, ns) == (len(mc), len(ms)) and ns > 500
        assert np.array_equal(bits(dev.map_cloud(0)), bits(mc)) and np.array_equal(bits(dev.map_cloud(1)), bits(ms))
    q = rng.uniform(-6, 6, (500, 3)).astype(np.float32)
    idx, d2 = m.nearestKSearch(1, q, 50.0)
    oi, od = orc.KdTree(ms).knn(q, 5)
    assert np.array_equal(np.where(od < 50.0, oi, -1), idx)
    dev.close(); m.close()


@pytest.mark.gpu
def test_device_mapping_loop_cell_mode_matches_oracle(gpu_lib, small_world):
    from loam_livox_amd.mapping import Laser_mapping
    from tests.test_mapping_sequence import MAP_ARGS, N_PTS, make_sequence
    scans, truth = make_sequence(small_world["world"], n_frames=7)
    args = dict(MAP_ARGS, matching_mode=1, maximum_in_fov_angle=50.0, threshold_cell_revisit=2000)
    om = LaserMapping(**args)
    lm = Laser_mapping(scan_points=N_PTS, cell_map_max_points=1 << 18, **args)
    for k, xyzi in enumerate(scans):
        ro = om.process_new_scan(xyzi)
        rd = lm.process_new_scan(xyzi)
        dt, dr = synth.pose_error(lm.pose, om.pose)
        assert rd == ro == 1 and dt < 1e-7 and dr < 1e-7
        assert lm.map_sizes == (len(om.maps[0]), len(om.maps[1]))
        assert lm.last_report.n_blocks_last == om.report.n_blocks_last
        if dt == 0.0 and dr == 0.0:
            assert np.array_equal(bits(lm.history.map_cloud(1)), bits(om.maps[1]))
    dt, dr = synth.pose_error(lm.pose, truth[len(scans) - 1])
    assert dt < 0.03 and dr < 0.006 and lm.map_sizes[1] > 300
    lm.close()


@pytest.mark.gpu
def test_append_touched_with_a_short_buffer_truncates_and_does_not_fail(gpu_lib):
    """ll_cellmap_append_touched stores the cloud before it lists the touched cells: a list buffer that is too short must not turn
    the call into an error -- a caller that retried would append the cloud twice.  The full count comes back, the list is cut."""
    import ctypes as C
    from loam_livox_amd import capi
    from loam_livox_amd.api import Cell_map
    rng = np.random.default_rng(12)
    cloud = np.c_[rng.uniform(-5, 5, (3000, 3)), np.zeros(3000)].astype(np.float32)
    full, short = Cell_map(1 << 14, 1.0), Cell_map(1 << 14, 1.0)
    want = full.append_cloud_touched(cloud, 3)
    assert len(want) > 40
    ijk = np.full((8, 3), -777, np.int32)
    n = C.c_int64(0)
    rc = short.L.ll_cellmap_append_touched(short.h, cloud.ctypes.data_as(C.c_void_p), len(cloud), 3, ijk.ctypes.data_as(C.c_void_p), 8, C.byref(n))
    assert rc == 0 and n.value == len(want)            # the full count, no error
    assert np.array_equal(ijk, want[:8])               # the first capacity_cells entries
    assert short.stats() == full.stats()               # the cloud is in exactly once
    full.close(); short.close()


@pytest.mark.gpu
def test_borrowed_cell_map_handle_settles_the_service_thread_on_every_read(gpu_lib):
    """a handle borrowed from ll_history_cell_map is kept across ll_history_add* while the history feeds its cell maps on
    the service thread -- which also grows them (cellmap_grow frees and swaps every array).  Every ll_cellmap_* entry point now waits for
    the frames handed over so far: stats / dump / device_view on the OLD handle, straight after the adds and without any explicit sync,
    see every frame; and a history without cell maps says so instead of returning a bare NULL."""
    from loam_livox_amd.api import History_buffer
    from loam_livox_amd.capi import LoamLivoxError
    rng = np.random.default_rng(11)
    frames = [(rng.uniform(-20, 20, (300, 4)).astype(np.float32), rng.uniform(-20, 20, (2500, 4)).astype(np.float32)) for _ in range(24)]
    pose = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)
    out = []
    for async_ in (True, False):
        h = History_buffer(maximum_history_size=30, max_points_per_frame=4000, line_res=0.05, plane_res=0.05)
        if async_:
            with pytest.raises(LoamLivoxError, match="not enabled"):
                h.cell_map(1)
        h.enable_cell_map(max_points=4000, cell_resolution=1.0)   # one frame's worth: the maps double several times on the way
        h.set_cell_map_async(async_)
        borrowed = [h.cell_map(0), h.cell_map(1)]                 # taken BEFORE any frame is in
        seen = []
        for k, (c, s) in enumerate(frames):
            h.add(c, s, pose)
            if k % 5 == 4:
                seen.append(borrowed[1].stats()[:2])              # no sync: the call itself settles the service thread
        dumps = [b.dump() for b in borrowed]
        out.append((seen, dumps))
        assert seen[-1][1] > 20000 and borrowed[1].stats()[1] == len(dumps[1][0])
        h.close()
    assert out[0][0] == out[1][0]
    for k in (0, 1):
        assert all(np.array_equal(a, b) for a, b in zip(out[0][1][k], out[1][1][k]))
