import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from loam_livox_amd import synth, capi
from loam_livox_amd.api import Livox_laser, Map_buffer, Point_cloud_registration
L = capi.load()
L.ll_reg_debug_cycles.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
world, corner, surf = synth.make_maps(5_000_000)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
base = [synth.make_scan(world, k) for k in range(8)]
scans = np.stack([base[b % 8].xyzi for b in range(B)]); init = np.stack([base[b % 8].pose_init for b in range(B)])
mp = Map_buffer(); mp.setInputCloud(0, corner); mp.setInputCloud(1, surf)
fe = Livox_laser(max_points=24000, max_scans=B, piecewise_number=1); fe.upload(scans, np.full(B, 1.0))
fe.extract_batch(B); fe.resolve(); fe.select_batch(B, -1, 0.0, 1.0)
reg = Point_cloud_registration(max_scans=B, max_features=24000)
p = reg.params; p.icp_max_iterations, p.ceres_max_iterations, p.force_all_iterations = 10, 20, 1
p.para_max_angular_rate, p.para_max_speed, p.max_final_cost = 20.0, 0.3, 1000.0; p.current_frame_index, p.mapping_init_accumulate_frames = 100, 50
for rep in range(2):
    res, pc, pi, reps = reg.solve_batch_fe(mp, fe, B, init, init)
tot = np.zeros(6)
for b in range(B):
    out = (C.c_longlong * 6)(); L.ll_reg_debug_cycles(reg.h, b, out); tot += np.array(list(out), float)
tot /= B
names = ["eval", "lm_ctl", "l1", "dedupe", "select+prune", "total"]
print("B=%d per scan (10 ICP iters), shader cycles: " % B + ", ".join("%s %.0fk (%.0f%%)" % (n, v / 1e3, 100 * v / tot[5]) for n, v in zip(names, tot)), "lm iters", np.mean([r.lm_iterations_total for r in reps]))
