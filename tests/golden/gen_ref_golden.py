#!/usr/bin/env python3
"""Generates tests/golden/ref_scene*.npz from the REFERENCE ITSELF: oracle/_ref/libll_ref.so, i.e. /root/reference's
livox_feature_extractor.hpp, ceres_icp.hpp and point_cloud_registration.hpp compiled verbatim in this container
against the stand-in third-party headers of oracle/ref_stubs/ (`make -C oracle ref`).

These fixtures are what travels: /root/reference does not exist on the GPU box, the committed vectors do.  They pin
  * the oracle      (tests/test_ref_golden.py, CPU tier), and
  * the HIP path    (same file, -m gpu)
to outputs of the reference's own code.  Feature-extraction fields are pure reference arithmetic (+ Eigen's 3-vector
dot / norm order); the registration outputs additionally depend on the stand-in FLANN / Ceres (see oracle/README.md).

Run from the repo root (needs /root/reference):  python tests/golden/gen_ref_golden.py
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loam_livox_amd import synth  # noqa: E402
from oracle import orc, ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
MAP_POINTS, MAP_SEED = 30_000, 777


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def main():
    assert ref.can_build(), "needs /root/reference"
    world, corner, surf = synth.make_maps(MAP_POINTS, seed=MAP_SEED)
    cases = [("mid40_a", synth.make_scan(world, 500, n=4000), 0), ("mid40_b", synth.make_scan(world, 501, n=4000), 0),
             ("moving", synth.make_moving_scan(world, 502, n=4000), 1)]
    for k, (name, sc, deblur) in enumerate(cases):
        xyzi = sc.xyzi.copy()
        if k == 1:  # exercise the masks: zero run, NaNs, duplicates, x == 0
            xyzi[100:110, :3] = 0
            xyzi[900, :3] = np.nan
            xyzi[901, 1] = np.inf
            xyzi[1500:1504] = xyzi[1500]
            xyzi[2000, 0] = 0
        R = ref.RefLivoxLaser()
        stamp = 1.0
        n_clouds = R.extract(xyzi, stamp)
        info = R.pts_info()
        pet = R.petals(n_clouds)
        out = dict(xyzi=xyzi, stamp=stamp, current_time=R.current_time(), n_petal_clouds=n_clouds,
                   petal_first=np.array([p[1][0] for p in pet], np.int32), petal_last=np.array([p[1][-1] for p in pet], np.int32),
                   petal_sizes=np.array([len(p[1]) for p in pet], np.int32))
        for f in ("pt_type", "pt_label", "time_stamp", "polar_angle", "polar_direction", "polar_dis_sq2", "depth_sq2", "curvature",
                  "view_angle", "sigma", "img2d"):
            out[f] = info[f]
        for tag, (lo, hi) in (("all", (0.0, 1.0)), ("w03", (0.0, 0.3)), ("mid", (0.35, 0.7))):
            g = R.get_features(lo, hi)
            out[f"{tag}_window"] = np.array([lo, hi], np.float32)
            out[f"{tag}_corners"] = g["pc_corners"]
            out[f"{tag}_surface"] = g["pc_surface"]
            out[f"{tag}_full"] = g["pc_full"]
            out[f"{tag}_corner_idx"] = g["corner_idx"]
            out[f"{tag}_surf_idx"] = g["surf_idx"]
        # registration with the reference's own driver (PCR:163-583), launch-file bounds, its own convergence break
        g = R.get_features(0.0, 1.0)
        prm = orc.RegParams.defaults(icp_iters=5, ceres_iters=20, deblur=deblur)
        if deblur:
            prm.minimum_pt_time_stamp = float(info["time_stamp"][0])
            prm.maximum_pt_time_stamp = float(info["time_stamp"][-1])
        RR = ref.RefRegistration()
        RR.set_params(prm)
        RR.set_maps(corner, surf)
        ret, pc, pi, rep = RR.solve(g["pc_corners"], g["pc_surface"], sc.pose_init, sc.pose_init)
        out.update(map_points=MAP_POINTS, map_seed=MAP_SEED, corner_crc=crc(corner), surf_crc=crc(surf), pose_init=sc.pose_init, deblur=deblur,
                   icp_iters=5, ceres_iters=20, t_min=prm.minimum_pt_time_stamp, t_max=prm.maximum_pt_time_stamp, reg_ret=ret, pose_out=pc,
                   pose_incre=pi, final_cost=rep["final_cost"], initial_cost=rep["initial_cost"], inlier_threshold=rep["inlier_threshold"],
                   n_blocks_last=rep["n_blocks_last"], angular_diff_deg=rep["angular_diff_deg"], t_diff=rep["t_diff"])
        np.savez_compressed(os.path.join(HERE, f"ref_scene{k}.npz"), **out)
        print(name, "petal clouds", n_clouds, "features", len(g["corner_idx"]), len(g["surf_idx"]), "ret", ret, "blocks", rep["n_blocks_last"],
              "pose err vs truth", synth.pose_error(pc, sc.pose_true))


if __name__ == "__main__":
    main()
