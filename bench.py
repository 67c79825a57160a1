#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on one node: scans/s of the scan-to-map hot path.

Workload (BASELINE config C2): 24 000-point synthetic Livox Mid-40 scans against a fixed 5 M-point
corner+surface map, 10 ICP iterations (convergence break disabled), k = 5, fp32 points; every extracted
feature is a query ("Q-full": maximum_residual_blocks >= feature count, no random sub-sampling).

One "step" = one pass of the hot path over one batch of B independent scans that are already resident in HBM:
feature extraction (K1-K4) -> selection (K3) -> per ICP iteration {transform + 5-NN + block build (K6),
prerun solve + inlier prune + final solve (K8/K9)} -> accept/reject.  Scans are independent units (the
reference's maximum_parallel_thread model, laser_mapping.hpp:1737-1742), so N GPUs run N independent shards
with no data-path collective ("weak" scaling); value = scans processed by all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="independent scans per step and per GPU")
    ap.add_argument("--map-points", type=int, default=5_000_000)
    ap.add_argument("--scan-points", type=int, default=24000)
    ap.add_argument("--icp-iters", type=int, default=10)
    ap.add_argument("--distinct-scans", type=int, default=16)
    ap.add_argument("--cell-corner", type=float, default=0.0, help="grid cell size override (0 = library default)")
    ap.add_argument("--cell-surf", type=float, default=0.0)
    ap.add_argument("--q-pipe", action="store_true", help="Q-pipe query mode (SURVEY 8d): device VoxelGrid (leaf 0.1 corner / 0.4 surface, "
                    "laser_mapping.hpp:742-743,1367-1373) between extraction and registration; default is Q-full")
    ap.add_argument("--force-general", action="store_true", help="A/B: run the HBM-resident solver path that large scans use")
    ap.add_argument("--legacy-solver", action="store_true", help="A/B: round-1 solver fast path (49-byte fp64 plane blocks, no LDS block cache)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-q-pipe", action="store_true", help="skip the secondary Q-pipe figure (profiling runs: keeps one launch shape per kernel)")
    ap.add_argument("--cpu-scans", type=int, default=16, help="scans of the step also run through the CPU oracle (about 10 s on one core)")
    return ap.parse_args()


def pmc_traffic_bytes(kernel: str, batch: int):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/r*_pmc_hbm_bytes.csv:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command at batch 256; FETCH_SIZE x2 per the
    gfx950 note in MI355X_MICROARCH.md).  None when no matching measurement is committed."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_bytes.csv")))
    if not files or batch != 256:
        return None
    best = None
    with open(files[-1]) as f:
        rows = [r for r in csv.reader(l for l in f if not l.startswith("#"))]
    hdr, rows = rows[0], rows[1:]
    for r in rows:
        d = dict(zip(hdr, r))
        if d["kernel"].split("<")[0] == "ll::" + kernel.replace("reg_knn_build_kernel", "reg_knn_kernel"):
            g = int(d["grid_threads"])
            if best is None or g > best[0]:
                best = (g, (2.0 * float(d["fetch_kib_avg"]) + float(d["write_kib_avg"])) * 1024.0)
    return None if best is None else int(best[1])


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch

    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist_mod.init_process_group(backend="nccl", rank=rank, world_size=world)
        dist = dist_mod
    dev = local_rank if torch.cuda.is_available() else 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU implementation)")
    torch.cuda.set_device(dev)

    from loam_livox_amd import synth
    from loam_livox_amd.api import Livox_laser, Map_buffer, Point_cloud_registration, VoxelGrid

    B, N = args.batch, args.scan_points
    t0 = time.time()
    world_model, corner, surf = synth.make_maps(args.map_points)
    n_distinct = min(args.distinct_scans, B)
    base = [synth.make_scan(world_model, 100 * rank + k, n=N) for k in range(n_distinct)]
    rng = np.random.default_rng(4242 + rank)
    scans = np.stack([base[b % n_distinct].xyzi for b in range(B)])
    poses_true = np.stack([base[b % n_distinct].pose_true for b in range(B)])
    # every slot gets its own initial-guess perturbation: distinct work per slot
    init = np.stack([
        synth.pose_compose(poses_true[b], np.r_[synth.quat_from_axis_angle(rng.normal(size=3), np.deg2rad(rng.uniform(0, 1.0))),
                                                 rng.uniform(-0.1, 0.1, 3)]) for b in range(B)])
    t_data = time.time() - t0

    mp = Map_buffer(device=dev)
    t0 = time.time()
    mp.setInputCloud(Map_buffer.CORNER, corner, args.cell_corner)
    mp.setInputCloud(Map_buffer.SURF, surf, args.cell_surf)
    t_map = time.time() - t0
    fe = Livox_laser(max_points=N, max_scans=B, device=dev, piecewise_number=1)
    fe.upload(scans, np.full(B, 1.0))
    reg = Point_cloud_registration(max_scans=B, max_features=N, device=dev)
    p = reg.params
    p.icp_max_iterations, p.ceres_max_iterations, p.force_all_iterations = args.icp_iters, 20, 1
    p.para_max_angular_rate, p.para_max_speed, p.max_final_cost = 20.0, 0.3, 1000.0
    p.current_frame_index, p.mapping_init_accumulate_frames = 100, 50
    p.maximum_allow_residual_block = N
    reg.set_profiling(True)
    if args.force_general or args.legacy_solver:
        reg.set_debug(False, force_general_solver=args.force_general, legacy_solver=args.legacy_solver)

    vox = (VoxelGrid(N, B, device=dev), VoxelGrid(N, B, device=dev)) if args.q_pipe else None

    def step():
        fe.extract_batch(B)
        fe.resolve()
        fe.select_batch(B, -1, 0.0, 1.0)
        if vox:
            reg.enqueue_fe_downsampled(mp, fe, vox[0], vox[1], 0.1, 0.4, B, init, init)
        else:
            reg.enqueue_fe(mp, fe, B, init, init)
        return reg.collect(B)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    k_ms = np.zeros(3)
    k_n = np.zeros(3)
    out = None
    for _ in range(args.steps):
        out = step()
        ms, n = reg.kernel_times()
        k_ms += ms
        k_n += n
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    res, pc, pi, reps = out
    nc, ns, nf, n_amb = fe.counts(B)
    nc_fe, ns_fe = nc, ns
    if vox:  # what the registrar saw
        nc, ns = vox[0].counts(B)[0], vox[1].counts(B)[0]
    total_scans = B * args.steps * world
    value = total_scans / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # single-scan latency (batch of 1 through the same code path)
    lat = []
    fe1 = Livox_laser(max_points=N, max_scans=1, device=dev, piecewise_number=1)
    fe1.upload(scans[:1], np.full(1, 1.0))
    reg1 = Point_cloud_registration(max_scans=1, max_features=N, device=dev)
    for f_ in ("icp_max_iterations", "ceres_max_iterations", "force_all_iterations", "para_max_angular_rate", "para_max_speed",
               "max_final_cost", "current_frame_index", "mapping_init_accumulate_frames", "maximum_allow_residual_block"):
        setattr(reg1.params, f_, getattr(p, f_))
    vox1 = (VoxelGrid(N, 1, device=dev), VoxelGrid(N, 1, device=dev)) if vox else None
    for i in range(5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fe1.extract_batch(1); fe1.resolve(); fe1.select_batch(1, -1, 0.0, 1.0)
        if vox:
            reg1.enqueue_fe_downsampled(mp, fe1, vox1[0], vox1[1], 0.1, 0.4, 1, init[:1], init[:1])
        else:
            reg1.enqueue_fe(mp, fe1, 1, init[:1], init[:1])
        reg1.collect(1)
        lat.append(time.perf_counter() - t1)
    latency_ms = 1e3 * float(np.median(lat[1:]))

    # secondary figure: the same workload in the other query mode (SURVEY 8d reports both), a short untimed-warm-up run
    q_pipe_extra = None
    if not vox and not args.no_q_pipe:
        vq = (VoxelGrid(N, B, device=dev), VoxelGrid(N, B, device=dev))

        def step_q():
            fe.extract_batch(B); fe.resolve(); fe.select_batch(B, -1, 0.0, 1.0)
            reg.enqueue_fe_downsampled(mp, fe, vq[0], vq[1], 0.1, 0.4, B, init, init)
            return reg.collect(B)

        step_q()
        barrier()
        tq = time.perf_counter()
        for _ in range(3):
            step_q()
        barrier()
        tq = (time.perf_counter() - tq) / 3
        ncq, nsq = vq[0].counts(B)[0], vq[1].counts(B)[0]
        q_pipe_extra = {"scans_per_s_this_rank": round(B / tq, 1), "ms_per_step": round(1e3 * tq, 3),
                        "features_per_scan": {"corner": float(ncq.mean()), "surface": float(nsq.mean())},
                        "note": "device VoxelGrid (leaf 0.1 / 0.4 m, laser_mapping.hpp:1367-1373) between extraction and registration"}
        reg.enqueue_fe(mp, fe, B, init, init)  # leave the registrar in the state of the timed configuration
        reg.collect(B)

    # roofline of the dominant kernel (HIP events on the registrar's stream, see ll_reg_set_profiling)
    names = ["reg_knn_build_kernel", "reg_solve_kernel", "reg_finalize_kernel"]
    dom = int(np.argmax(k_ms))
    avg_ms = float(k_ms[dom] / max(1.0, k_n[dom]))
    line_blocks = float(sum(r.corner_avail for r in reps))
    plane_blocks = float(sum(r.surf_avail for r in reps))
    queries = float(nc.sum() + ns.sum())
    # residual-block constants as stored (data layout): 16 B point + 1 B flag + 48 B (line: a', u') or
    # 32 B (plane: n', n'.a')
    block_bytes = line_blocks * 65.0 + plane_blocks * 49.0
    if dom == 1:
        # algorithmic bytes of one solver launch: every residual block's constants read once + the 28 reduced doubles
        # per scan (DESIGN.md "roofline")
        alg_bytes = block_bytes + B * 224.0
    else:
        # k-NN + block build launch: 16 B/query in, the block constants out (candidate gather is cache-resident)
        alg_bytes = queries * 16.0 + block_bytes
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    traffic = pmc_traffic_bytes(names[dom], B)
    roofline = {"bound": "hbm", "kernel": names[dom], "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 6), "traffic": traffic, "avg_launch_ms": round(avg_ms, 4),
                "algorithmic_bytes_per_launch": int(alg_bytes),
                # what the kernel really moves (PMC, profiles/) against the same peak: how close the sweeps run to HBM speed
                "traffic_frac": None if traffic is None else round(traffic / (avg_ms * 1e-3) / 8e12, 4)}

    result = {
        "metric": "scans_per_s", "value": round(value, 2), "unit": "scans/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 points / f64 solve", "data": "synthetic",
        "config": {"workload": "C2: 24k-pt Mid-40 scan vs 5M-pt corner+surface map, 10 ICP iters (fixed), k=5, "
                               + ("Q-pipe (device VoxelGrid 0.1/0.4 before registration)" if vox else "Q-full"),
                   "scan_points": N, "map_points": int(len(corner) + len(surf)), "map_corner": int(len(corner)),
                   "map_surf": int(len(surf)), "icp_iters": args.icp_iters, "batch_scans_per_step_per_gpu": B,
                   "parallelism": f"replicas x{world} (independent scans, no data-path collective)"},
        "roofline": roofline,
        "q_pipe": q_pipe_extra,
        "kernel_ms_per_step": {names[i]: round(float(k_ms[i] / args.steps), 3) for i in range(3)},
        "per_iter_knn_jtj_ms_per_batch": round(float((k_ms[0] + k_ms[1]) / args.steps / max(1, args.icp_iters)), 4),
        "single_scan_latency_ms": round(latency_ms, 3),
        "features_per_scan": {"corner": float(nc.mean()), "surface": float(ns.mean())},
        "knn_reuse_last_iter": dict(zip(("searched", "resorted"), reg.debug_worklists(B)), queries=int(nc.sum() + ns.sum()),
                                    corner_searched_resorted=reg.debug_worklists_by_kind(B)[0], corner_queries=int(nc.sum())),
        "solver_phase_cycles_scan0": [int(v) for v in reg.debug_cycles(0)],
        "accepted_frac": float(np.mean(res)), "lm_iters_per_scan": float(np.mean([r.lm_iterations_total for r in reps])),
        "ambiguous_labels": int(n_amb), "setup_s": {"synthetic_data": round(t_data, 1), "map_upload_grid_build": round(t_map, 2)},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # the CPU leg belongs to the N = 1 line only
        # CPU baseline: the oracle (C restatement of the reference algorithm; the reference binary cannot be built
        # here) on one host core, bounded sample of the same workload.  It is the checker, never the product.
        from oracle import orc
        tb = time.perf_counter()
        tc, ts = orc.KdTree(corner), orc.KdTree(surf)
        t_tree = time.perf_counter() - tb
        prm = orc.RegParams.defaults(icp_iters=args.icp_iters, ceres_iters=20, force_all=1)
        prm.max_final_cost = 1000.0
        t_cpu, errs, same_sets, same_lm, same_blocks, same_res = 0.0, [], True, True, True, True
        n_cpu = min(args.cpu_scans, B)
        for b in range(n_cpu):
            tb = time.perf_counter()
            o = orc.fe_extract(scans[b], 1.0)
            ci, si, fi = orc.fe_get_features(o, 0.0, 1.0)
            fc_o, fs_o = orc.feature_cloud(o, ci), orc.feature_cloud(o, si)
            if vox:
                fc_o, fs_o = orc.voxel_grid(fc_o, 0.1)[1], orc.voxel_grid(fs_o, 0.4)[1]
            ret, opc, _, orep = orc.reg_solve(tc, ts, fc_o, fs_o, prm, init[b], init[b])
            t_cpu += time.perf_counter() - tb
            errs.append(synth.pose_error(pc[b], opc))
            same_sets &= (len(ci) == nc_fe[b] and len(si) == ns_fe[b] and len(fc_o) == nc[b] and len(fs_o) == ns[b])
            same_lm &= (orep.lm_iterations_total == reps[b].lm_iterations_total and orep.icp_iterations == reps[b].icp_iterations)
            same_blocks &= (orep.n_blocks_last == reps[b].n_blocks_last and orep.corner_avail == reps[b].corner_avail and orep.surf_avail == reps[b].surf_avail)
            same_res &= (ret == res[b])
        result["cpu_baseline"] = {"value": round(n_cpu / t_cpu, 4), "unit": "scans/s", "cores": 1, "kind": "port",
                                  "sample": f"{n_cpu} of the {B} scans of one step (extract + {args.icp_iters} ICP iters each) "
                                            f"vs the same {len(corner) + len(surf)}-pt map; k-d tree build {t_tree:.1f}s excluded",
                                  "host_cores_available": os.cpu_count()}
        result["parity_vs_cpu"] = {"max_pose_err_m": float(max(e[0] for e in errs)), "max_pose_err_rad": float(max(e[1] for e in errs)),
                                   "feature_counts_identical": bool(same_sets), "lm_and_icp_iteration_counts_identical": bool(same_lm),
                                   "residual_block_counts_identical": bool(same_blocks), "accept_reject_identical": bool(same_res),
                                   "scans_compared": int(n_cpu)}
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
