"""Writes tests/golden/ref_cells*.npz: what the REFERENCE'S OWN cell map and key-frame classes (source/cell_map_keyframe.hpp compiled
verbatim into oracle/_ref/libll_ref_cells.so, oracle/ref_cells.py) produce on seeded clouds.  Runs only where /root/reference exists;
the fixtures travel to the GPU box, the library's sources do not.

    python tests/golden/gen_ref_cells.py

Per fixture: the clouds; after every append the cells of cell_vec and m_current_frame_idx; the final store (cells, points in (cell,
insertion) order, m_last_update_frame_idx); find_cells_in_radius answers; determine_feature of every cell; Maps_keyframe::analyze over
(a) every cell of the map, (b) the cells a second map gets from the first cloud alone, (c) the cells cell_vec named after the first
cloud -- the way the mapping loop fills a key frame (laser_mapping.hpp:1529-1562); max_similiarity_of_two_image of (a) against (b)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_cells as rc  # noqa: E402


def clouds(seed, offset, n_frames, sizes=(1500, 300, 300)):
    """planes, lines and scattered points around `offset`; one cloud is shifted away (its cells are new), one is tiny, one is empty"""
    rng = np.random.default_rng(seed)
    frames = []
    for f in range(n_frames):
        o = rng.uniform(-4, 4, 3) * np.array([1, 1, 0.3])
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        u = np.cross(n, [0, 0, 1.0])
        u /= np.linalg.norm(u)
        v = np.cross(n, u)
        plane = o + rng.uniform(-2, 2, (sizes[0], 1)) * u + rng.uniform(-2, 2, (sizes[0], 1)) * v + rng.normal(0, 0.01, (sizes[0], 3))
        line = o + n * 0.5 + rng.uniform(-2, 2, (sizes[1], 1)) * u + rng.normal(0, 0.01, (sizes[1], 3))
        blob = rng.uniform(-5, 5, (sizes[2], 3))
        c = np.concatenate([plane, line, blob])
        if f == 3:
            c[:, 0] += 12.0
        if f == 4:
            c = c[:2]
        if f == 5:
            c = c[:0]
        c = (c + np.asarray(offset)).astype(np.float32)
        frames.append(np.c_[c, np.full(len(c), 7.0, np.float32)].astype(np.float32))
    return frames


def ragged(rows, width, dtype):
    off = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    flat = np.concatenate([np.asarray(r, dtype).reshape(-1, width) for r in rows]) if off[-1] else np.zeros((0, width), dtype)
    return flat, off


def analysis(m, ijk):
    kf = rc.RefKeyframe()
    kf.add_cells(m, ijk)
    a = kf.analyze()
    kf.close()
    return a


def fixture(name, seed, offset, n_frames, resolution, revisit):
    frames = clouds(seed, offset, n_frames)
    m = rc.RefCellMap(resolution, revisit)
    touched, frame_idx = [], []
    for c in frames:
        touched.append(m.append(c))
        frame_idx.append(m.frame_idx())
    ijk, cnt, last = m.cells()
    pts = [m.cell_points(k) for k in ijk]
    assert all(len(p) == n for p, n in zip(pts, cnt))
    store_xyz = np.concatenate(pts).astype(np.float32)
    start = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    feats = [m.feature(k) for k in ijk]
    cov6 = np.array([[f["cov"][0, 0], f["cov"][0, 1], f["cov"][0, 2], f["cov"][1, 1], f["cov"][1, 2], f["cov"][2, 2]] for f in feats], np.float32)
    rng = np.random.default_rng(seed + 1)
    rad_q = np.c_[(rng.uniform(-6, 6, (6, 3)) + np.asarray(offset)).astype(np.float32), np.array([0.4, 1.0, 2.5, 5.0, 9.0, 40.0], np.float32)]
    rad_cells, rad_off = ragged([m.cells_in_radius(q[:3], q[3]) for q in rad_q], 3, np.int32)
    t_flat, t_off = ragged(touched, 3, np.int32)
    kf_all = analysis(m, ijk)
    m0 = rc.RefCellMap(resolution)
    m0.append(frames[0])
    kf_first = analysis(m0, m0.cells()[0])
    later = np.unique(np.concatenate([t for t in touched[1:] if len(t)]), axis=0)
    kf_later = analysis(m, later)
    sim_plane = rc.max_similarity(kf_all["images"][1], kf_first["images"][1])
    sim_line = rc.max_similarity(kf_all["images"][0], kf_first["images"][0])
    out = dict(resolution=np.float32(resolution), revisit=np.int64(revisit), frame_sizes=np.array([len(c) for c in frames]),
               frames=np.concatenate(frames), touched=t_flat, touched_off=t_off, frame_idx=np.array(frame_idx, np.int32),
               cell_ijk=ijk.astype(np.int32), cell_start=start, cell_last=last.astype(np.int32), store_xyz=store_xyz,
               radius_query=rad_q, radius_cells=rad_cells, radius_off=rad_off,
               feat_type=np.array([f["type"] for f in feats], np.int32), feat_vector=np.array([f["vector"] for f in feats], np.float32),
               feat_mean=np.array([f["mean"] for f in feats], np.float32), feat_cov=cov6,
               feat_eval=np.array([f["eigen_val"] for f in feats], np.float32), later_cells=later.astype(np.int32),
               sim_plane=np.float64(sim_plane), sim_line=np.float64(sim_line))
    for tag, a in (("all", kf_all), ("first", kf_first), ("later", kf_later)):
        for k, v in a.items():
            out[f"kf_{tag}_{k}"] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, len(ijk), "cells,", len(store_xyz), "points, labels", np.bincount(out["feat_type"], minlength=3), "frame idx", frame_idx,
          "touched", [len(t) for t in touched], "n_vec", kf_all["n_vectors"], kf_later["n_vectors"], "sim", sim_plane, sim_line)
    m.close()
    m0.close()


if __name__ == "__main__":
    if not rc.available():
        sys.exit("the reference library is not built (no /root/reference here)")
    fixture("ref_cells0.npz", 4711, (0.0, 0.0, 0.0), 5, 1.0, 2)
    fixture("ref_cells1.npz", 99, (180.0, -140.0, 20.0), 7, 0.8, 2**31 - 1)
