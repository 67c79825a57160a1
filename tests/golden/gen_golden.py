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


if __name__ == "__main__":
    main()
