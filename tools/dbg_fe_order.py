#!/usr/bin/env python3
"""Order dependence: tests/test_ref_golden.py's GPU test, then tests/test_ref_c2.py's steps with diagnostics."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from loam_livox_amd import capi, synth  # noqa: E402
from loam_livox_amd.api import Livox_laser, Map_buffer  # noqa: E402
import tests.test_ref_golden as tg  # noqa: E402
import tests.test_ref_c2 as tc  # noqa: E402

L = capi.load()
which = sys.argv[1] if len(sys.argv) > 1 else "golden"
if which == "golden":
    for p in tg.SCENES:
        tg.test_hip_path_reproduces_reference_outputs(L, p)
    print("golden GPU test body done")
world, corner, surf = synth.make_maps(5_000_000)
m = Map_buffer()
if which != "nomap":
    m.setInputCloud(Map_buffer.CORNER, corner)
    m.setInputCloud(Map_buffer.SURF, surf)
for path in tc.SCENES[:2]:
    g = np.load(path)
    sc = tc.scan_of(world, g)
    n = len(sc.xyzi)
    for rep in range(2):
        fe = Livox_laser(max_points=n, max_scans=1, piecewise_number=1)
        nc = fe.extract_laser_features(sc.xyzi, float(g["stamp"]))
        info = fe.pts_info()
        print(os.path.basename(path), "rep", rep, "petal clouds", nc, "expected", int(g["n_petal_clouds"]), "labels nonzero", int(np.count_nonzero(info["pt_label"])),
              "type nonzero", int(np.count_nonzero(info["pt_type"])), "time first/last", info["time_stamp"][0], info["time_stamp"][-1],
              "depth nan", int(np.isnan(info["depth_sq2"]).sum()), "polar max", float(np.nanmax(info["polar_dis_sq2"])))
        fe.close()
