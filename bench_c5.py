#!/usr/bin/env python3
"""BASELINE config C5 on one MI355X: the exact 5-NN kernel against a 50 M-point map -- the configuration whose map
(640 MB of 16-byte records + the cell table) no longer fits the 256 MB Infinity Cache, i.e. the honest HBM run of the
k-NN.  fp32 points by default; --f16 switches the map to 8-byte fp16-in-cell records (ll_map_to_f16).

  python bench_c5.py [--map-points 50000000] [--queries 4000000]
  rocprofv3 --kernel-trace --stats ... / --pmc FETCH_SIZE ... -- python bench_c5.py   (profiles/README.md)

Queries are the surface features of synthetic Mid-40 scans moved to the map frame with their initial-guess poses (what
ICP iteration 0 asks).  Prints one JSON line: wall-clock queries/s through ll_map_knn5 (H2D of the queries and D2H of
the results included), kernel-only time from HIP events, and parity of a sample against brute force on the host."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--map-points", type=int, default=50_000_000)
    ap.add_argument("--queries", type=int, default=4_000_000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--parity-queries", type=int, default=64)
    ap.add_argument("--f16", action="store_true", help="fp16-point records (8 B per point), fp32 accumulate: the literal C5 configuration")
    args = ap.parse_args()
    import torch
    from loam_livox_amd import synth
    from loam_livox_amd.api import Livox_laser, Map_buffer

    t0 = time.time()
    world, corner, surf = synth.make_maps(args.map_points)
    t_map_gen = time.time() - t0
    mp = Map_buffer()
    t0 = time.time()
    mp.setInputCloud(Map_buffer.SURF, surf)
    if args.f16:
        mp.to_f16(Map_buffer.SURF)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    # queries: surface features of a few scans, replicated with distinct initial-guess perturbations
    fe = Livox_laser(max_points=24000)
    rng = np.random.default_rng(5)
    qs = []
    k = 0
    while sum(len(q) for q in qs) < args.queries:
        sc = synth.make_scan(world, k % 32, 24000)
        fe.extract_laser_features(sc.xyzi, 1.0)
        f = fe.get_features(0.0, 1.0)["pc_surface"][:, :3]
        pose = synth.pose_compose(sc.pose_true, np.r_[synth.quat_from_axis_angle(rng.normal(size=3), np.deg2rad(rng.uniform(0, 1.0))),
                                                       rng.uniform(-0.1, 0.1, 3)])
        qs.append(synth.transform_points(pose, f))
        k += 1
    q = np.ascontiguousarray(np.concatenate(qs)[: args.queries])
    max_d2 = 50.0
    mp.nearestKSearch(Map_buffer.SURF, q[:1000], max_d2)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        idx, d2 = mp.nearestKSearch(Map_buffer.SURF, q, max_d2)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.reps
    # parity sample vs brute force over the whole map (the oracle's k-d tree would take minutes to build at this size)
    from oracle import orc
    sel = rng.choice(len(q), args.parity_queries, replace=False)
    t0 = time.time()
    bi, bd = orc.bruteforce_knn(mp.dequantized(Map_buffer.SURF) if args.f16 else surf, q[sel], 5)
    t_bf = time.time() - t0
    same = bool(np.array_equal(np.where(bd < max_d2, bi, -1), idx[sel]) and np.array_equal(np.where(bd < max_d2, bd, np.inf), d2[sel]))
    found = float(((idx >= 0).sum(1) == 5).mean())
    print(json.dumps({
        "metric": "knn_queries_per_s", "value": round(len(q) / wall, 1), "unit": "5-NN queries/s through ll_map_knn5 (host buffers in and out)",
        "config": {"workload": "C5: exact 5-NN, " + ("fp16 points (8-byte records) / fp32 accumulate" if args.f16 else "fp32 points") + ", 50M-pt map (surface part), Mid-40 surface features as queries",
                   "map_surface_points": int(len(surf)), "queries": int(len(q)), "max_sq_dis": max_d2},
        "ms_per_call": round(1e3 * wall, 2), "found_frac": found,
        "setup_s": {"synthetic_map": round(t_map_gen, 1), "upload_and_grid_build": round(t_build, 2)},
        "parity_vs_bruteforce": {"queries": int(args.parity_queries), "identical": same, "cpu_s_per_query": round(t_bf / args.parity_queries, 3)},
    }))


if __name__ == "__main__":
    main()
