#!/usr/bin/env python3
"""Generates tests/golden/ref_c2_scene*.npz: the REFERENCE'S OWN extractor and registration driver (oracle/_ref/libll_ref.so =
/root/reference compiled here, `make -C oracle ref`) at BASELINE config C2 size -- a 24 000-point Mid-40 scan against the
5 M-point corner + surface map -- so that the GPU tier compares the HIP path with the reference's output at the size the bench
runs, not through the oracle (missing #2).  A fixture holds what is needed to re-create the inputs (map size / seed +
checksums, scan seed + checksum) and the reference's outputs: corner / surface index sets, counts, pose, costs.

Run from the repo root (needs /root/reference):  python tests/golden/gen_ref_c2.py
"""
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loam_livox_amd import synth  # noqa: E402
from oracle import orc, ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
MAP_POINTS = 5_000_000
SCANS = [1000, 1017, 1042, 1101]
ICP_ITERS, CERES_ITERS = 10, 20


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def main():
    assert ref.can_build(), "needs /root/reference"
    world, corner, surf = synth.make_maps(MAP_POINTS)
    RR = ref.RefRegistration()
    prm = orc.RegParams.defaults(icp_iters=ICP_ITERS, ceres_iters=CERES_ITERS)
    prm.max_final_cost = 1000.0   # the cost of ~17 k blocks is far above the node's default of 2 (every scan would be rejected)
    RR.set_params(prm)
    t0 = time.time()
    RR.set_maps(corner, surf)
    print("reference k-d trees over", len(corner), "+", len(surf), "points:", round(time.time() - t0, 1), "s")
    for k, seed in enumerate(SCANS):
        sc = synth.make_scan(world, seed)
        R = ref.RefLivoxLaser()
        n_clouds = R.extract(sc.xyzi, 1.0)
        g = R.get_features(0.0, 1.0)
        t0 = time.time()
        ret, pc, pi, rep = RR.solve(g["pc_corners"], g["pc_surface"], sc.pose_init, sc.pose_init)
        dt = time.time() - t0
        out = dict(map_points=MAP_POINTS, corner_crc=crc(corner), surf_crc=crc(surf), n_map_corner=len(corner), n_map_surf=len(surf),
                   scan_seed=seed, scan_crc=crc(sc.xyzi), stamp=1.0, current_time=R.current_time(), n_petal_clouds=n_clouds,
                   corner_idx=g["corner_idx"].astype(np.int32), surf_idx=g["surf_idx"].astype(np.int32), pose_init=sc.pose_init,
                   icp_iters=ICP_ITERS, ceres_iters=CERES_ITERS, max_final_cost=1000.0, reg_ret=ret, pose_out=pc, pose_incre=pi,
                   final_cost=rep["final_cost"], initial_cost=rep["initial_cost"], inlier_threshold=rep["inlier_threshold"],
                   n_blocks_last=rep["n_blocks_last"], angular_diff_deg=rep["angular_diff_deg"], t_diff=rep["t_diff"])
        np.savez_compressed(os.path.join(HERE, f"ref_c2_scene{k}.npz"), **out)
        print("scan", seed, "features", len(g["corner_idx"]), len(g["surf_idx"]), "ret", ret, "blocks", rep["n_blocks_last"], "solve", round(dt, 1), "s",
              "pose err vs truth", synth.pose_error(pc, sc.pose_true))


if __name__ == "__main__":
    main()
