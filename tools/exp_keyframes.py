#!/usr/bin/env python3
"""exploration: what the key frames of the out-and-back sequence of tests/test_keyframes.py look like to the loop detector
(cells, non-zero ratios, similarities, gates, alignment)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loam_livox_amd import synth
from loam_livox_amd.keyframes import Keyframe_assembly
from loam_livox_amd.api import keyframe_similarity

world = synth.world_for_map_size(200_000)
rng = np.random.default_rng(77)
start = synth.sensor_pose_in_world(world, rng)
ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)
zax, yax = np.array([0.0, 0.0, 1.0]), np.array([0.0, 1.0, 0.0])
per_kf = 40
for icp_iters, thr in ((4, 0.35), (10, 0.35)):
    ka = Keyframe_assembly(scans_of_each_keyframe=per_kf, scans_between_two_keyframe=per_kf, minimum_keyframe_differen=2,
                           maximum_keyframe_in_waiting_list=3, map_alignment_inlier_threshold=thr, map_alignment_maximum_icp_iteration=icp_iters,
                           max_points=1 << 22, avail_ratio_plane=0.02, avail_ratio_line=0.0)
    away = synth.pose_compose(start, np.r_[synth.quat_from_axis_angle(zax, np.deg2rad(170.0)), np.array([12.0, 6.0, 0.0])])
    drift = np.r_[synth.quat_from_axis_angle(zax, np.deg2rad(0.5)), np.array([0.6, -0.4, 0.1])]
    k = 0
    for grp, (base, err, span, pitch, jit) in enumerate([(start, ident, 220.0, 15.0, 0.0), (away, ident, 300.0, 20.0, 0.0), (start, drift, 360.0, 30.0, 0.5)]):
        for j in range(per_kf):
            yaw, pit = np.deg2rad(span * (j / (per_kf - 1) - 0.5)), np.deg2rad(pitch * np.sin(3.1 * j))
            rot = synth.quat_mul(synth.quat_from_axis_angle(zax, yaw), synth.quat_from_axis_angle(yax, pit))
            true_pose = synth.pose_compose(base, np.r_[rot, jit * np.array([np.sin(1.7 * j), np.cos(2.3 * j), 0.0])])
            sc = synth.make_moving_scan(world, 9100 + 100 * grp + j, 24000, inc_true=ident, pose_start=true_pose, t_phase=0.07 * j)
            est = synth.pose_compose(err, true_pose)
            ok = np.isfinite(sc.xyzi[:, :3]).all(axis=1) & (np.abs(sc.xyzi[:, :3]).sum(axis=1) > 0)
            cloud = np.c_[synth.transform_points(est, sc.xyzi[ok, :3]), np.zeros(int(ok.sum()), np.float32)].astype(np.float32)
            k += 1
            ka.add_scan(cloud, est, k)
            found = ka.process_waiting()
            if found:
                print("  LOOP", {kk: (np.round(v, 3).tolist() if isinstance(v, np.ndarray) else v) for kk, v in found[0].items()})
    print(f"icp iterations {icp_iters}: key frames {len(ka.keyframe_vec)}; drift {drift[4:7].tolist()}")
    for i, kf in enumerate(ka.keyframe_vec):
        a = kf.analysis
        print(f"  kf{i}: cells {len(kf.m_set_cell)} ratio_nonzero {np.round(a['ratio_nonzero'], 4).tolist()} n_vectors {a['n_vectors'].tolist()} roi_range {a['roi_range']:.2f}")
    print("  log", [{kk: (round(v, 3) if isinstance(v, float) else (np.round(v, 3).tolist() if isinstance(v, np.ndarray) else v)) for kk, v in r.items()} for r in ka.log])
    ka.close()
