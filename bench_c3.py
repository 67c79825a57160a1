#!/usr/bin/env python3
"""bench_c3.py -- BASELINE config C3 data point (not the driver's bench.py contract): Mid-100 = three Mid-40 heads
(3 x 24 000 points, yaw -38.4 / 0 / +38.4 deg) taken from a MOVING sensor, registered with motion-distortion
compensation (if_motion_deblur = 1, the *_mb residuals) against a 20 M-point map.  The timed step is extract + register:
the 3 B raw head scans are resident in HBM, every step extracts their features (one extractor batch of 3 B slots, every
head its own time base), merges the three heads of a sweep on the device (ll_reg_enqueue_fe_merged,
laser_feature_extractor.hpp:348-358) and registers; the merged scan has > 24 576 residual blocks, so the registrar takes
its general (HBM-resident) solver path.  --features-resident times the registration alone on pre-merged clouds (the
round-1 figure).  Prints one JSON line."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


# The ROCm runtime multiplexes a process's HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  Batches in flight live on their
# own handles = their own streams; with 4 queues two of them regularly share one and their kernels serialise (measured, profiles/README.md round 5:
# 44.0 k scans/s with 4 queues, 46.5 k with 12 - 32; Q-pipe with three 2 048-scan batches in flight 122 k -> 169 k).  Must be set before the
# first HIP call of the process; an explicit setting of the caller wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--map-points", type=int, default=20_000_000)
    ap.add_argument("--distinct", type=int, default=4)
    ap.add_argument("--icp-iters", type=int, default=10)
    ap.add_argument("--cpu-scans", type=int, default=1)
    ap.add_argument("--features-resident", action="store_true", help="time the registration only, merged feature clouds uploaded once (round-1 mode)")
    ap.add_argument("--in-flight", type=int, default=2, help="batches in flight in the timed loop (1 = one at a time)")
    ap.add_argument("--no-deblur", action="store_true", help="A/B: register the same scans without the motion-deblur residuals")
    args = ap.parse_args()
    from loam_livox_amd import synth
    from loam_livox_amd.api import Livox_laser, Map_buffer, Point_cloud_registration

    world, corner, surf = synth.make_maps(args.map_points)
    N, B = 24000, args.batch
    fe = Livox_laser(max_points=N, piecewise_number=1)
    merged, raw = [], []
    for k in range(args.distinct):
        rng = np.random.default_rng(7000 + k)
        pose_start = synth.sensor_pose_in_world(world, rng)
        inc = np.r_[synth.quat_from_axis_angle(rng.normal(size=3), np.deg2rad(rng.uniform(0.3, 1.0))), rng.uniform(-0.08, 0.08, 3)]
        cs, ss = [], []
        for h, yaw in enumerate(np.deg2rad([-38.4, 0.0, 38.4])):
            sc = synth.make_moving_scan(world, 10 * k + h, n=N, inc_true=inc, yaw_offset=float(yaw), pose_start=pose_start)
            fe2 = Livox_laser(max_points=N, piecewise_number=1)   # each head has its own time base (stamp 0 -> t = i*1e-5)
            fe2.upload(sc.xyzi[None], np.array([0.0]))
            fe2.extract_batch(1); fe2.resolve()
            g = fe2.get_features(0.0, 1.0)
            cs.append(g["pc_corners"]); ss.append(g["pc_surface"])
            raw.append(sc.xyzi)
            fe2.close()
        merged.append((np.concatenate(cs), np.concatenate(ss), pose_start, synth.pose_compose(pose_start, inc)))
    corners = [merged[b % args.distinct][0] for b in range(B)]
    surfs = [merged[b % args.distinct][1] for b in range(B)]
    pose_last = np.stack([merged[b % args.distinct][2] for b in range(B)])
    pose_true = np.stack([merged[b % args.distinct][3] for b in range(B)])
    nfeat = max(max(len(c) for c in corners), max(len(s) for s in surfs))

    mp = Map_buffer()
    t0 = time.time()
    mp.setInputCloud(Map_buffer.CORNER, corner); mp.setInputCloud(Map_buffer.SURF, surf)
    t_map = time.time() - t0
    def make_slot():
        """a registrar and the step that feeds it (its own extractor handle, or its own copy of the merged feature clouds)"""
        r_ = Point_cloud_registration(max_scans=B, max_features=nfeat)
        q_ = r_.params
        q_.if_motion_deblur, q_.minimum_pt_time_stamp, q_.maximum_pt_time_stamp = (0 if args.no_deblur else 1), 0.0, float((N - 1) * np.float32(1e-5))
        q_.icp_max_iterations, q_.ceres_max_iterations, q_.force_all_iterations = args.icp_iters, 20, 1
        q_.para_max_angular_rate, q_.para_max_speed, q_.max_final_cost = 20.0, 0.3, 1000.0
        q_.current_frame_index, q_.mapping_init_accumulate_frames = 100, 50
        q_.maximum_allow_residual_block = 3 * N
        r_.set_profiling(True)
        if args.features_resident:
            # the merged feature clouds are uploaded once: the timed region starts with them resident in HBM
            r_.upload_features(corners, surfs)

            def step_():
                r_.enqueue_uploaded(mp, B, pose_last, pose_last)
        else:
            # the raw head scans are uploaded once; every step extracts, merges the heads on the device and registers
            f_ = Livox_laser(max_points=N, max_scans=3 * B, piecewise_number=1)
            f_.upload(np.stack([raw[3 * (b % args.distinct) + h] for b in range(B) for h in range(3)]), np.zeros(3 * B))
            f_.sync()

            def step_():
                f_.extract_batch(3 * B)
                f_.resolve()
                f_.select_batch(3 * B, -1, 0.0, 1.0)
                r_.enqueue_fe_merged(mp, f_, B, 3, pose_last, pose_last)
        return r_, step_

    t0 = time.perf_counter()
    reg, step = make_slot()
    t_upload = time.perf_counter() - t0
    for _ in range(args.warmup):
        step()
        reg.collect(B)
    t0 = time.perf_counter()
    kms, kn = np.zeros(3), np.zeros(3)
    for _ in range(args.steps):
        step()
        res, pc, pi, reps = reg.collect(B)
        kms += reg.kernel_times()[0]
        kn += reg.kernel_times()[1]
    el = time.perf_counter() - t0
    sequential = None
    if args.in_flight > 1:
        # the same steps with consecutive batches overlapped (bench.py does the same): batch i+1 is extracted / enqueued on its own
        # handles and streams while batch i's kernels run; same work inside the timed region, same results
        slots = [(reg, step)] + [make_slot() for _ in range(args.in_flight - 1)]
        D = len(slots)

        def pipelined(k_steps):
            o = None
            for j in range(min(D - 1, k_steps)):
                slots[j][1]()
            for i in range(k_steps):
                if i + D - 1 < k_steps:
                    slots[(i + D - 1) % D][1]()
                o = slots[i % D][0].collect(B)
            return o

        pipelined(D)
        t0 = time.perf_counter()
        out_p = pipelined(args.steps)
        el_p = time.perf_counter() - t0
        sequential = {"value": round(B * args.steps / el, 2), "ms_per_step": round(1e3 * el / args.steps, 2),
                      "results_equal_bitwise": bool(np.array_equal(out_p[1], pc) and np.array_equal(out_p[0], res))}
        el = el_p
    err = [synth.pose_error(pc[b], pose_true[b]) for b in range(B)]
    out = {"metric": "scans_per_s", "value": round(B * args.steps / el, 2), "unit": ("Mid-100 scans/s (registration with deblur; merged feature clouds resident in HBM, their one-off upload is reported as feature_upload_s)"
                    if args.features_resident else "Mid-100 scans/s (extract 3 heads + merge on the device + register with deblur; raw scans resident in HBM)"),
           "config": {"workload": "C3: 3x24k-pt Mid-100 scan from a moving sensor vs 20M-pt map, if_motion_deblur=1, 10 ICP iters (fixed)",
                      "map_points": int(len(corner) + len(surf)), "batch": B, "features_per_scan": {"corner": float(np.mean([len(c) for c in corners])), "surface": float(np.mean([len(s) for s in surfs]))}},
           "ms_per_step": round(1e3 * el / args.steps, 2), "batches_in_flight": max(1, args.in_flight), "one_batch_at_a_time": sequential, "kernel_ms_per_step": {"knn+build": round(float(kms[0] / args.steps), 2), "solver": round(float(kms[1] / args.steps), 2)},
           "accepted_frac": float(np.mean(res)), "median_err_vs_truth_m": float(np.median([e[0] for e in err])), "median_err_vs_truth_rad": float(np.median([e[1] for e in err])),
           "blocks_last": float(np.mean([r.n_blocks_last for r in reps])), "map_upload_grid_build_s": round(t_map, 2), "feature_upload_s" if args.features_resident else "scan_upload_s": round(t_upload, 3)}
    # ---- roofline of the dominant kernel: the plane-table solver of large / motion-deblur scans (reg_solve_big_kernel<1>, ll_reg_big_path.h), one
    #      launch per ICP iteration; algorithmic bytes (definition unchanged since round 1) = every residual block's constants once per
    #      launch (49 B per plane block: fp32 point + blur ratio, {a0, v0}, {v1, v2}, flag; 65 B per line block) + 224 B of state out per scan;
    #      duration from the HIP events around the solver launches on the registrar's stream (one batch at a time)
    n_line = float(np.sum([r.corner_avail for r in reps])), float(np.sum([r.surf_avail for r in reps]))
    alg = n_line[0] * 65.0 + n_line[1] * 49.0 + 224.0 * B
    ms_launch = float(kms[1] / max(1.0, kn[1]))
    traffic = src = None
    import csv, glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_c3_pmc_hbm_bytes.csv")))
    if hits:
        best = None
        for r in csv.DictReader(l for l in open(hits[-1]) if not l.startswith("#")):
            if "reg_solve_big_kernel" in r["kernel"] and (best is None or int(r["grid_threads"]) > int(best["grid_threads"])):
                best = r
        if best is not None and int(best["grid_threads"]) == B * 512:
            traffic, src = int(float(best["fetch_kib_avg"]) * 2048 + float(best["write_kib_avg"]) * 1024), os.path.relpath(hits[-1], ROOT)
    out["roofline"] = {"bound": "hbm", "kernel": "reg_solve_big_kernel<1>", "achieved": round(alg / (ms_launch * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                       "frac": round(alg / (ms_launch * 1e-3) / 1e9 / 8000.0, 4), "avg_launch_ms": round(ms_launch, 4), "launches_timed": int(kn[1]),
                       "algorithmic_bytes_per_launch": int(alg), "traffic": traffic, "traffic_source": src,
                       "traffic_over_algorithmic": (round(traffic / alg, 2) if traffic else None),
                       "hbm_gb_per_s_from_counters": (round(traffic / (ms_launch * 1e-3) / 1e9, 1) if traffic else None),
                       "limited_by": "fp64 VALU issue of the motion-deblur evaluations (the interpolated rotation, its Jacobian and the 27 accumulator updates per "
                                     "block: ~250 wave instructions per block and evaluation, two wavefronts per SIMD), then the per-launch census / hash "
                                     "de-duplication of the neighbour triples and the five L1 sweeps of the inlier threshold -- not HBM bandwidth"}
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from build_id import build_id, read_stamp
    if src:
        bid, commit = read_stamp(os.path.join(ROOT, src))
        out["roofline"].update({"traffic_source_build": bid, "traffic_source_commit": commit, "build": build_id(), "traffic_is_current": (bid == build_id()) if bid else None})
    cyc = np.array([[int(v) for v in reg.debug_cycles(b)] for b in range(B)], np.float64)
    if cyc.any():  # a -DLL_SOLVE_TIMING build (LOAM_LIVOX_LIB=...): shader-clock cycles per scan, summed over the registration's launches
        out["solver_phase_cycles_mean_over_scans"] = [int(v) for v in cyc.mean(0)]
        out["solver_phase_cycles_of_the_slowest_scan"] = [int(v) for v in cyc[int(np.argmax(cyc[:, 5]))]]
        out["solver_phase_cycles_first_scans"] = [[int(v) for v in cyc[b]] for b in range(min(B, args.distinct))]
        out["solver_phase_cycles_legend"] = "0 evaluations, 1 controller, 2 L1 pass, 3 set de-duplication, 4 rank select (+ prune), 5 whole solver call, 6.. path specific (ll_device.h RegState::dbg_cycles)"
    if args.cpu_scans > 0:
        from oracle import orc
        tb = time.perf_counter()
        tc, ts = orc.KdTree(corner), orc.KdTree(surf)
        t_tree = time.perf_counter() - tb
        prm = orc.RegParams.defaults(icp_iters=args.icp_iters, ceres_iters=20, force_all=1, deblur=1)
        prm.minimum_pt_time_stamp, prm.maximum_pt_time_stamp, prm.max_final_cost = 0.0, float((N - 1) * np.float32(1e-5)), 1000.0
        tb = time.perf_counter()
        ret, opc, _, orep = orc.reg_solve(tc, ts, corners[0], surfs[0], prm, pose_last[0], pose_last[0])
        t_cpu = time.perf_counter() - tb
        dt, dr = synth.pose_error(pc[0], opc)
        out["cpu_baseline"] = {"value": round(1.0 / t_cpu, 4), "unit": "scans/s", "cores": 1, "kind": "port", "sample": f"1 merged scan; k-d tree build {t_tree:.1f}s excluded"}
        out["parity_vs_cpu"] = {"pose_err_m": dt, "pose_err_rad": dr, "blocks_equal": bool(orep.n_blocks_last == reps[0].n_blocks_last)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
