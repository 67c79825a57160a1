#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle on seeded synthetic inputs.

The reference ships no golden vectors and cannot be built here (PCL/Ceres/Eigen/ROS absent), so these fixtures are
ORACLE-generated ("parity unpinned", see oracle/ll_oracle.h): they freeze the oracle's answers so that (a) any change
to the oracle is visible in review and (b) the HIP path is checked against committed numbers as well as against a
live oracle run.  Run from the repo root:  python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loam_livox_amd import synth  # noqa: E402
from oracle import orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    world, corner, surf = synth.make_maps(30_000, seed=777)
    for k in (0, 1):
        sc = synth.make_scan(world, 500 + k, n=4000)
        fe = orc.fe_extract(sc.xyzi, 2.0)
        S, first, last = orc.fe_split_scan(fe)
        ps, pe = orc.fe_piecewise(fe, first, last, 2)
        ci, si, fi = orc.fe_get_features(fe, 0.0, 1.0)
        fc, fs = orc.feature_cloud(fe, ci), orc.feature_cloud(fe, si)
        tc, ts = orc.KdTree(corner), orc.KdTree(surf)
        qs = synth.transform_points(sc.pose_init, fs[:, :3])
        knn_i, knn_d = ts.knn(qs, 5)
        prm = orc.RegParams.defaults(icp_iters=4, ceres_iters=20, force_all=1)
        ret, pc, pi, rep = orc.reg_solve(tc, ts, fc, fs, prm, sc.pose_init, sc.pose_init)
        np.savez_compressed(
            os.path.join(HERE, f"scene{k}.npz"), xyzi=sc.xyzi, corner_map=corner, surf_map=surf, pose_init=sc.pose_init,
            current_time=2.0, pt_type=fe.pt_type, pt_label=fe.pt_label, depth_sq2=fe.depth_sq2, curvature=fe.curvature,
            split_idx=fe.split_idx, n_petals=fe.n_petals, petal_first=first, petal_last=last, piece_start=ps, piece_end=pe,
            corner_idx=ci, surf_idx=si, full_idx=fi, knn_surf_idx=knn_i, knn_surf_d2=knn_d, reg_ret=ret, pose_out=pc, pose_incre=pi,
            final_cost=rep.final_cost, n_blocks_last=rep.n_blocks_last, lm_iterations_total=rep.lm_iterations_total,
            inlier_threshold=rep.inlier_threshold)
        print("scene", k, "features", len(ci), len(si), "blocks", rep.n_blocks_last, "ret", ret)


def cells():
    """cell-map path: appends with the revisit rule, radius + field-of-view query with replace, labels, key-frame images"""
    from oracle.orc_cellmap import CellMap
    rng = np.random.default_rng(4711)
    frames = []
    for f in range(5):
        o = rng.uniform(-4, 4, 3) * np.array([1, 1, 0.3])
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        u = np.cross(n, [0, 0, 1.0]); u /= np.linalg.norm(u); v = np.cross(n, u)
        plane = o + rng.uniform(-2, 2, (1500, 1)) * u + rng.uniform(-2, 2, (1500, 1)) * v + rng.normal(0, 0.01, (1500, 3))
        line = o + n * 0.5 + rng.uniform(-2, 2, (300, 1)) * u + rng.normal(0, 0.01, (300, 3))
        blob = rng.uniform(-5, 5, (300, 3))
        c = np.concatenate([plane, line, blob]).astype(np.float32)
        if f == 3:
            c[:, 0] += 12.0
        frames.append(np.c_[c, np.full(len(c), 7.0, np.float32)].astype(np.float32))
    pose = np.r_[synth.quat_from_axis_angle(np.array([0.1, -0.2, 1.0]), np.deg2rad(20.0)), [-6.0, 0.5, 0.2]]
    m = CellMap(1.0, 2)
    for c in frames:
        m.append(c)
    cat, keys = m.query_filter(pose, 9.0, 50.0, 0.2, 1)
    xyz, ijk, start, last = m.dump()
    f = m.features()
    kf = m.keyframe_images(0.9)
    other = CellMap(1.0)
    other.append(frames[0])
    sim = CellMap.max_similarity(kf["images"][1], other.keyframe_images(0.0)["images"][1])
    np.savez_compressed(
        os.path.join(HERE, "cells0.npz"), frames=np.stack(frames), pose=pose, radius=9.0, fov=50.0, leaf=0.2, revisit=2, resolution=1.0,
        query_cloud=cat, n_selected=len(keys), store_xyz=xyz, cell_ijk=ijk, cell_start=start, cell_last=last, feat_type=f["type"],
        feat_vector=f["vector"], feat_mean=f["mean"], feat_cov=f["cov"], feat_eval=f["eigen_val"], feat_margin=f["margin"],
        kf_images=kf["images"], kf_ratio=kf["ratio_nonzero"], kf_R=kf["eigen_R"], kf_nvec=kf["n_vectors"], kf_centre=kf["centre"],
        kf_range=kf["roi_range"], kf_near_edge=kf["near_bin_edge"], similarity=sim)
    print("cells0:", len(m.cells), "cells,", m.n_points(), "points,", len(keys), "selected,", np.bincount(f["type"], minlength=3), "labels, sim", sim)


if __name__ == "__main__":
    main()
    cells()
