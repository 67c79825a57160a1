#!/usr/bin/env python3
"""BASELINE config C5 on one MI355X: the exact 5-NN search against a 50 M-point map as the HBM-bound run it is specified to be
(BASELINE.json configs[4]: "fp16 points / fp32 accumulate k-NN, 50M-pt map, rocprof HBM-GB/s roofline run").

The map's surface part (~40 M points: 640 MB of 16-byte records, or 320 MB of fp16-in-cell records, + the cell table) does not fit the
256 MB Infinity Cache -- but only a run whose QUERIES cover the map reads it from HBM (round 4's 32 scan poses touched 52 MB of it and
ran out of cache).  Here the queries are the map's own surface points, jittered by a few centimetres and grouped into "scans" of
--scan-queries spatially compact points (points sorted by an 8 m coarse cell, consecutive chunks): thousands of scan positions spread
over the whole world, every map cell queried, U = the map.

Three searches, queries and results RESIDENT on the device (nothing crosses PCIe inside a timed call):
  per_lane_fp32 / per_lane_fp16   ll_map_knn5_device: one lane per query (knn5_kernel / knn5_f16_kernel), kernel time from HIP events on
                                  the map's stream (inside the library: torch events would watch the wrong stream);
  tile_fp32                       the registrar's default path for large batches (reg_qsort_kernel + reg_knn_tile_kernel, ll_knn_tile.h)
                                  on the same queries as B scans, one ICP iteration; k-NN class time from ll_reg_set_profiling's events.
`value` = queries/s of the fp16 per-lane search (the literal C5 configuration) unless --headline says otherwise.  roofline.achieved =
ALGORITHMIC bytes / kernel time, algorithmic = the map once (records + 4-byte cell starts) + 12 B per query in + 40 B per query out
(20 B of indices, 20 B of squared distances; the registrar path writes a 16-byte neighbour record + a flag instead); roofline.traffic
= FETCH_SIZE x 2 + WRITE_SIZE of the same kernel from profiles/r05_c5_pmc_hbm_bytes.csv when present (tools/gpu_r5_c3c5_prof.sh).
Parity: a sample of every search against brute force over the whole map on the host."""
import argparse
import csv
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def pmc_bytes(kernel_substr):
    """(fetch x 2 + write) bytes per dispatch of the largest-grid instance of a kernel, from the newest committed C5 counter summary"""
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_c5_pmc_hbm_bytes.csv")))
    if not hits:
        return None, None
    best = None
    for r in csv.DictReader(l for l in open(hits[-1]) if not l.startswith("#")):
        if kernel_substr in r["kernel"] and (best is None or int(r["grid_threads"]) > int(best["grid_threads"])):
            best = r
    if best is None:
        return None, None
    return int(float(best["fetch_kib_avg"]) * 1024 * 2 + float(best["write_kib_avg"]) * 1024), os.path.relpath(hits[-1], ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--map-points", type=int, default=50_000_000)
    ap.add_argument("--scan-queries", type=int, default=16384, help="queries per synthetic scan position (a spatially compact chunk of the map)")
    ap.add_argument("--max-scans", type=int, default=0, help="use only the first this many scan positions (0 = all: the queries cover the map)")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--parity-queries", type=int, default=32)
    ap.add_argument("--jitter", type=float, default=0.03)
    ap.add_argument("--only", default="", help="comma list of per_lane_fp32,tile_fp32,per_lane_fp16 (profiling runs: one kernel shape per run)")
    ap.add_argument("--headline", default="per_lane_fp32", help="the run `value` quotes (round 6: the fp16-record variant is slower than fp32 -- a gather costs a cache line whatever the record size -- and is reported beside it, not as the headline)")
    ap.add_argument("--query-order", default="scan", choices=["scan", "cell"],
                    help="scan: the chunks' points in map-array order (default); cell: every scan's queries sorted by the 0.6 m map cell they fall into, x fastest (experiment: what a cell-ordered query stream gives the per-lane search)")
    args = ap.parse_args()
    import torch
    from loam_livox_amd import synth
    from loam_livox_amd.api import Map_buffer, Point_cloud_registration
    from oracle import orc

    t0 = time.time()
    world, corner, surf = synth.make_maps(args.map_points)
    t_map_gen = time.time() - t0
    n_map = len(surf)
    # queries = the map's points, jittered, in scans of spatially compact chunks
    rng = np.random.default_rng(5)
    cell = np.floor(surf[:, :3] / 8.0).astype(np.int64)
    cell -= cell.min(0)
    key = (cell[:, 2] * (cell[:, 1].max() + 1) + cell[:, 1]) * (cell[:, 0].max() + 1) + cell[:, 0]
    order = np.argsort(key, kind="stable")
    Q = args.scan_queries
    B = n_map // Q
    if args.max_scans > 0:
        B = min(B, args.max_scans)
    sel = order[: B * Q]
    q = (surf[sel, :3] + rng.normal(0.0, args.jitter, (B * Q, 3))).astype(np.float32)
    if args.query_order == "cell":
        c = np.floor((q - q.min(0)) / np.float32(0.6)).astype(np.int64)
        k2 = (c[:, 2] * (c[:, 1].max() + 1) + c[:, 1]) * (c[:, 0].max() + 1) + c[:, 0]
        k2 += (np.arange(B * Q) // Q) * (k2.max() + 1)   # inside each scan
        q = q[np.argsort(k2, kind="stable")]
        del c, k2
    del cell, key, order
    max_d2 = 50.0
    want = [w for w in args.only.split(",") if w] or ["per_lane_fp32", "tile_fp32", "per_lane_fp16"]
    out_runs, parity = {}, {}
    psel = rng.choice(len(q), args.parity_queries, replace=False)

    mp = Map_buffer()
    t0 = time.time()
    mp.setInputCloud(Map_buffer.SURF, surf)
    mp.setInputCloud(Map_buffer.CORNER, corner)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    n_cells = mp.cells(Map_buffer.SURF)
    dq = torch.from_numpy(q).cuda()
    d_idx = torch.empty((len(q), 5), dtype=torch.int32, device="cuda")
    d_d2 = torch.empty((len(q), 5), dtype=torch.float32, device="cuda")

    def roofline(kernel, ms, rec_bytes, out_bytes_per_query, extra=None):
        alg = n_map * rec_bytes + 4 * (n_cells + 1) + len(q) * (12 + out_bytes_per_query)
        tr, src = pmc_bytes(kernel)
        r = {"bound": "hbm", "kernel": kernel, "achieved": round(alg / (ms * 1e-3) / 1e9, 1), "peak": PEAK_GBS, "unit": "GB/s",
             "frac": round(alg / (ms * 1e-3) / 1e9 / PEAK_GBS, 4), "avg_launch_ms": round(ms, 3), "algorithmic_bytes_per_launch": int(alg),
             "algorithmic": f"map once ({n_map} records x {rec_bytes} B + {n_cells + 1} cell starts x 4 B) + {len(q)} queries x (12 B in + {out_bytes_per_query} B out)",
             "traffic": tr, "traffic_source": src, "traffic_over_algorithmic": (round(tr / alg, 3) if tr else None),
             "hbm_gb_per_s_from_counters": (round(tr / (ms * 1e-3) / 1e9, 1) if tr else None)}
        if extra:
            r.update(extra)
        return r

    def per_lane(tag, kernel, rec_bytes):
        mp.nearestKSearch_device(Map_buffer.SURF, dq[: 4 * Q], max_d2, d_idx[: 4 * Q], d_d2[: 4 * Q])  # warm-up
        ms = [mp.nearestKSearch_device(Map_buffer.SURF, dq, max_d2, d_idx, d_d2) for _ in range(args.reps)]
        ms = float(np.median(ms))
        idx, d2 = d_idx[torch.from_numpy(psel).cuda()].cpu().numpy(), d_d2[torch.from_numpy(psel).cuda()].cpu().numpy()
        found = float(((d_idx >= 0).sum(1) == 5).float().mean().item())
        out_runs[tag] = {"queries_per_s": round(len(q) / (ms * 1e-3), 1), "kernel_ms": round(ms, 3), "found_frac": found,
                         "roofline": roofline(kernel, ms, rec_bytes, 40)}
        return idx, d2

    def brute(points):
        t0 = time.time()
        bi, bd = orc.bruteforce_knn(points, q[psel], 5)
        return np.where(bd < max_d2, bi, -1), np.where(bd < max_d2, bd, np.inf), (time.time() - t0) / len(psel)

    bi32 = bd32 = None
    if "per_lane_fp32" in want or "tile_fp32" in want:
        bi32, bd32, t_bf = brute(surf)
    if "per_lane_fp32" in want:
        idx, d2 = per_lane("per_lane_fp32", "knn5_kernel", 16)
        parity["per_lane_fp32"] = bool(np.array_equal(bi32, idx) and np.array_equal(bd32, d2))

    if "tile_fp32" in want:
        reg = Point_cloud_registration(max_scans=B, max_features=Q)
        p = reg.params
        p.icp_max_iterations, p.ceres_max_iterations, p.ceres_prerun_times, p.force_all_iterations = 1, 1, 1, 1
        p.para_max_angular_rate, p.para_max_speed, p.max_final_cost = 20.0, 0.3, 1e9
        p.current_frame_index, p.mapping_init_accumulate_frames = 100, 50
        p.maximum_allow_residual_block = Q
        p.maximum_dis_plane_for_match = max_d2
        reg.set_profiling(True)
        reg.set_debug(True)
        feat = np.zeros((B, Q, 4), np.float32)
        feat[:, :, :3] = q.reshape(B, Q, 3)
        empty = np.zeros((0, 4), np.float32)
        reg.upload_features([empty] * B, list(feat))
        ident = np.tile(np.array([0, 0, 0, 1, 0, 0, 0], np.float64), (B, 1))
        ms_k = []
        for _ in range(args.reps + 1):
            reg.enqueue_uploaded(mp, B, ident, ident)
            reg.collect(B)
            ms_k.append(float(reg.kernel_times()[0][0]))
        ms = float(np.median(ms_k[1:]))
        # the lists the tile search stored, against brute force (through the registrar's debug tap)
        same = True
        for i in psel[: max(4, len(psel) // 4)]:
            b, k = int(i) // Q, int(i) % Q
            _, _, si, sd = reg.debug_knn(b, 0, Q)
            j = int(np.nonzero(psel == i)[0][0])
            full = bi32[j].min() >= 0
            same &= bool((not full) or (np.array_equal(si[k], bi32[j]) and np.array_equal(sd[k], bd32[j])))
        parity["tile_fp32"] = same
        out_runs["tile_fp32"] = {"queries_per_s": round(len(q) / (ms * 1e-3), 1), "knn_class_ms": round(ms, 3), "scans": B,
                                 "roofline": roofline("reg_knn_tile_kernel", ms, 16, 17, {"note": "k-NN class of one ICP iteration: query sort + tile search + block flags (HIP events, ll_reg_set_profiling)"})}
        reg.close()
        del feat

    if "per_lane_fp16" in want:
        mp.to_f16(Map_buffer.SURF)
        torch.cuda.synchronize()
        deq = mp.dequantized(Map_buffer.SURF)
        bi16, bd16, t_bf = brute(np.ascontiguousarray(deq, np.float32))
        idx, d2 = per_lane("per_lane_fp16", "knn5_f16_kernel", 12)  # 8-byte record + 4-byte original index
        parity["per_lane_fp16"] = bool(np.array_equal(bi16, idx) and np.array_equal(bd16, d2))

    head = args.headline if args.headline in out_runs else next(iter(out_runs))
    print(json.dumps({
        "metric": "knn_queries_per_s", "value": out_runs[head]["queries_per_s"], "unit": "5-NN queries/s, queries and results resident in HBM (kernel time from HIP events)",
        "headline_run": head, "dtype": "f16 points / f32 accumulate" if head.endswith("fp16") else "f32",
        "config": {"workload": "C5: exact 5-NN vs the 50M-pt map (surface part), queries = the map's own points jittered by 3 cm in spatially compact scans: the queries cover the map",
                   "map_surface_points": int(n_map), "map_cells": int(n_cells), "scan_positions": int(B), "queries": int(len(q)), "max_sq_dis": max_d2},
        "roofline": out_runs[head]["roofline"], "runs": out_runs,
        "parity_vs_bruteforce": {"queries_per_run": int(len(psel)), "identical": parity, "cpu_s_per_query": round(t_bf, 3)},
        "cpu_baseline": {"value": round(1.0 / t_bf, 2), "unit": "queries/s", "cores": 1, "kind": "port", "sample": f"{len(psel)} brute-force queries over the whole map (the oracle's k-d tree takes minutes to build at this size)"},
        "setup_s": {"synthetic_map": round(t_map_gen, 1), "upload_and_grid_build": round(t_build, 2)},
    }))


if __name__ == "__main__":
    main()
