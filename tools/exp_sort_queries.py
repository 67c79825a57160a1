#!/usr/bin/env python3
"""Experiment (k-NN lane utilisation): does ordering each scan's queries by map cell speed up the full 5-NN search?
Registers the same B scans twice through host-uploaded feature clouds -- in scan order and sorted by the 0.6 m cell of
their initial map position -- with 2 ICP iterations (the two full searches) and prints the k-NN class kernel time of each."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loam_livox_amd import synth  # noqa: E402
from loam_livox_amd.api import Livox_laser, Map_buffer, Point_cloud_registration  # noqa: E402

B, N = 128, 24000
world, corner, surf = synth.make_maps(5_000_000)
mp = Map_buffer()
mp.setInputCloud(Map_buffer.CORNER, corner)
mp.setInputCloud(Map_buffer.SURF, surf)
fe = Livox_laser(max_points=N, piecewise_number=1)
cs, ss, poses = [], [], []
for k in range(16):
    sc = synth.make_scan(world, 300 + k)
    fe.extract_laser_features(sc.xyzi, 1.0)
    g = fe.get_features(0.0, 1.0)
    cs.append(g["pc_corners"]); ss.append(g["pc_surface"]); poses.append(sc.pose_init)


def cell_order(f, pose, h):
    w = synth.transform_points(pose, f[:, :3])
    c = np.floor((w - w.min(0)) / h).astype(np.int64)
    return np.lexsort((c[:, 0], c[:, 1], c[:, 2]))


out = {}
for name in ("scan_order", "cell_order"):
    corners = [cs[b % 16] if name == "scan_order" else cs[b % 16][cell_order(cs[b % 16], poses[b % 16], 1.45)] for b in range(B)]
    surfs = [ss[b % 16] if name == "scan_order" else ss[b % 16][cell_order(ss[b % 16], poses[b % 16], 0.6)] for b in range(B)]
    pl = np.stack([poses[b % 16] for b in range(B)])
    reg = Point_cloud_registration(max_scans=B, max_features=N)
    p = reg.params
    p.icp_max_iterations, p.ceres_max_iterations, p.force_all_iterations = 2, 20, 1
    p.para_max_angular_rate, p.para_max_speed, p.max_final_cost = 20.0, 0.3, 1000.0
    p.current_frame_index, p.mapping_init_accumulate_frames = 100, 50
    reg.set_profiling(True)
    reg.upload_features(corners, surfs)
    for _ in range(2):
        reg.enqueue_uploaded(mp, B, pl, pl)
        res, pc, _, _ = reg.collect(B)
    reg.enqueue_uploaded(mp, B, pl, pl)
    res, pc, _, _ = reg.collect(B)
    ms, n = reg.kernel_times()
    out[name] = {"knn_class_ms_per_2_iterations": float(ms[0]), "solver_ms": float(ms[1]), "pose0": pc[0].tolist()}
    reg.close()
d = np.abs(np.array(out["scan_order"]["pose0"]) - np.array(out["cell_order"]["pose0"])).max()
out["pose_difference_between_orders"] = float(d)
print(json.dumps(out))
