#!/usr/bin/env python3
"""exploration: what the key frames of a short out-and-back sequence look like to the loop detector (ratios, similarities, alignment)"""
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
for per_kf, yaw_span in ((40, 300.0),):
    ka = Keyframe_assembly(scans_of_each_keyframe=per_kf, scans_between_two_keyframe=per_kf, minimum_keyframe_differen=2, maximum_keyframe_in_waiting_list=3,
                           map_alignment_inlier_threshold=0.35, map_alignment_maximum_icp_iteration=4, max_points=1 << 22)
    away = synth.pose_compose(start, np.r_[synth.quat_from_axis_angle(np.array([0.0, 0.0, 1.0]), np.deg2rad(170.0)), np.array([3.0, 1.0, 0.0])])
    drift = np.r_[synth.quat_from_axis_angle(np.array([0.0, 0.0, 1.0]), np.deg2rad(0.5)), np.array([0.6, -0.4, 0.1])]
    k = 0
    for grp, (base, err) in enumerate([(start, ident), (away, ident), (start, drift)]):
        for j in range(per_kf):
            yaw = np.deg2rad(yaw_span * (j / max(1, per_kf - 1) - 0.5))
            true_pose = synth.pose_compose(base, np.r_[synth.quat_from_axis_angle(np.array([0.0, 0.0, 1.0]), yaw), np.zeros(3)])
            sc = synth.make_moving_scan(world, 9100 + 100 * grp + j, 24000, inc_true=ident, pose_start=true_pose, t_phase=0.07 * j)
            est = synth.pose_compose(err, true_pose)
            ok = np.isfinite(sc.xyzi[:, :3]).all(axis=1) & (np.abs(sc.xyzi[:, :3]).sum(axis=1) > 0)
            cloud = np.c_[synth.transform_points(est, sc.xyzi[ok, :3]), np.zeros(int(ok.sum()), np.float32)].astype(np.float32)
            k += 1
            ka.add_scan(cloud, est, k)
            found = ka.process_waiting()
            if found:
                print("  LOOP", {kk: (np.round(v, 3).tolist() if isinstance(v, np.ndarray) else v) for kk, v in found[0].items()})
    print(f"per_kf {per_kf} yaw_span {yaw_span}: key frames {len(ka.keyframe_vec)}")
    for i, kf in enumerate(ka.keyframe_vec):
        a = kf.analysis
        print(f"  kf{i}: cells {len(kf.m_set_cell)} ratio_nonzero {np.round(a['ratio_nonzero'], 4).tolist()} n_vectors {a['n_vectors'].tolist()} roi_range {a['roi_range']:.2f}")
    for i in range(len(ka.keyframe_vec)):
        for j in range(i):
            a, b = ka.keyframe_vec[i], ka.keyframe_vec[j]
            print(f"  sim({i},{j}): plane {keyframe_similarity(a.m_feature_img_plane, b.m_feature_img_plane):.3f} line {keyframe_similarity(a.m_feature_img_line, b.m_feature_img_line):.3f}")
    print("  log", [{kk: (round(v, 3) if isinstance(v, float) else (np.round(v, 3).tolist() if isinstance(v, np.ndarray) else v)) for kk, v in r.items()} for r in ka.log])
    ka.close()
