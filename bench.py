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

The K timed steps keep three batches in flight (--in-flight): batch i+1 is extracted and its registration enqueued, on a second
extractor handle / registrar with their own streams, while batch i's kernels run -- all of every batch's work (extraction,
selection, the ICP iterations, the result download) lies inside the timed region, the results are bit-identical to processing
one batch at a time (the line says so: pipeline.results_equal_sequential_bitwise), and the one-at-a-time figure is reported
beside it (`sequential`; --no-pipeline makes it the value).

What the JSON line carries beyond the contract fields:
  value / ms_per_step     scans resident in HBM when the timed region starts (the contract's figure)
  sequential              the same K steps one batch at a time; kernel_ms_per_step and roofline are measured in THIS loop
                          (launches of two batches sharing the device stretch each other's event intervals)
  streamed                the same pipelined steps with every batch crossing PCIe inside the timed region: page-locked host
                          buffers, asynchronous copies on the extractor handle's stream
  q_pipe                  secondary figure: voxel-filtered queries (the node's input_downsample_mode), 4 batches in flight
  roofline                dominant kernel: algorithmic bytes per launch / average launch duration (HIP events), HBM peak
  roofline_path           SURVEY 8(d)'s whole-path figure  B_scan x scans/s / 8e12  with U, C counted per scan
                          (oracle/orc_roofline.py)
  cpu_baseline            the oracle (CPU restatement; gcc -O3) on all host cores, one scan per thread (the reference's
                          maximum_parallel_thread model); cpu_baseline_1thread: median of >= 20 single-scan runs after
                          3 warm-ups; cpu_baseline_shipped_config: the shipped operating point (Q-pipe, 200 blocks)

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The ROCm runtime multiplexes a process's HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  Batches in flight live on their
# own handles = their own streams; with 4 queues two of them regularly share one and their kernels serialise (measured, profiles/README.md round 5:
# 44.0 k scans/s with 4 queues, 46.5 k with 12 - 32; Q-pipe with three 2 048-scan batches in flight 122 k -> 169 k).  Must be set before the
# first HIP call of the process; an explicit setting of the caller wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="independent scans per step and per GPU")
    ap.add_argument("--map-points", type=int, default=5_000_000)
    ap.add_argument("--scan-points", type=int, default=24000)
    ap.add_argument("--icp-iters", type=int, default=10)
    ap.add_argument("--distinct-scans", type=int, default=256, help="distinct synthetic scans in the batch (each slot also gets its own initial guess)")
    ap.add_argument("--cell-corner", type=float, default=0.0, help="grid cell size override (0 = library default)")
    ap.add_argument("--cell-surf", type=float, default=0.0)
    ap.add_argument("--q-pipe", action="store_true", help="Q-pipe query mode (SURVEY 8d): device VoxelGrid (leaf 0.1 corner / 0.4 surface, "
                    "laser_mapping.hpp:742-743,1367-1373) between extraction and registration; default is Q-full")
    ap.add_argument("--force-general", action="store_true", help="A/B: run the HBM-resident solver path that large scans use")
    ap.add_argument("--no-knn-coop", action="store_true", help="A/B: corner searches one per lane everywhere (no wavefront-cooperative search)")
    ap.add_argument("--no-knn-tile", action="store_true", help="A/B: per-lane search of the surface queries + neighbour reuse (round 3) instead of the tile search")
    ap.add_argument("--knn-tile-with-reuse", action="store_true", help="A/B: tile search at ICP iterations 0 / 1, neighbour reuse + work lists afterwards")
    ap.add_argument("--no-solver-groups", action="store_true", help="A/B for the single-scan latency figure: one solver workgroup per scan "
                    "even for small batches (default: batches of <= 16 scans spread every scan over 8 workgroups)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / collective plumbing only (gloo, no GPU work): the CPU test of the --gpus path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-q-pipe", action="store_true", help="skip the secondary Q-pipe figure (profiling runs: keeps one launch shape per kernel)")
    ap.add_argument("--no-streamed", action="store_true", help="skip the PCIe-inclusive figure")
    ap.add_argument("--no-pipeline", action="store_true", help="value = one batch at a time (no overlap of consecutive batches)")
    ap.add_argument("--q-pipe-in-flight", type=int, default=3, help="batches in flight for the secondary Q-pipe figure")
    ap.add_argument("--q-pipe-batch", type=int, default=2048, help="scans per batch of the secondary Q-pipe figure (the small-scan solver packs several scans per CU)")
    ap.add_argument("--in-flight", type=int, default=3, help="batches in flight in the timed loop (extractor handle + registrar each)")
    ap.add_argument("--cpu-runs", type=int, default=20, help="timed single-thread oracle runs (after 3 warm-ups); their median is cpu_baseline_1thread")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the all-core CPU baseline (0 = os.cpu_count())")
    return ap.parse_args()


def newest_profile(pattern):
    """the newest committed summary of the DEFAULT line's passes (round tags sort lexicographically; the Q-pipe / C3 / C5 passes of a round carry
    their own infix and are not this command's)"""
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", pattern))
                   if not any(t in os.path.basename(f) for t in ("_qpipe_", "_c3_", "_c4_", "_c5_")))
    return files[-1] if files else None


def pmc_traffic_bytes(kernel: str, batch: int):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/r*_pmc_hbm_bytes.csv:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command at batch 256; FETCH_SIZE x2 per the
    gfx950 note in MI355X_MICROARCH.md).  (None, None) when no matching measurement is committed."""
    import csv
    path = newest_profile("r*_pmc_hbm_bytes.csv")
    if not path or batch != 256:
        return None, None
    best = None
    with open(path) as f:
        rows = [r for r in csv.reader(l for l in f if not l.startswith("#"))]
    hdr, rows = rows[0], rows[1:]
    for r in rows:
        d = dict(zip(hdr, r))
        if d["kernel"].split("<")[0] == "ll::" + kernel.replace("reg_knn_build_kernel", "reg_knn_tile_kernel"):
            g = int(d["grid_threads"])
            if best is None or g > best[0]:
                best = (g, (2.0 * float(d["fetch_kib_avg"]) + float(d["write_kib_avg"])) * 1024.0)
    return (None, None) if best is None else (int(best[1]), os.path.relpath(path, ROOT))


def traffic_stamp(rel_path):
    """which kernel sources the committed counter summary was taken on, and whether they are the ones running now (tools/build_id.py)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from build_id import build_id, read_stamp
    if not rel_path:
        return {}
    bid, commit = read_stamp(os.path.join(ROOT, rel_path))
    now = build_id()
    return {"traffic_source_build": bid, "traffic_source_commit": commit, "build": now,
            "traffic_is_current": (bid == now) if bid else None}  # None: a summary from before the stamps (round <= 5)


def pmc_valu_fp64(kernel: str, batch: int):
    """fp64 VALU work per launch of `kernel` from the newest committed instruction-mix summary (profiles/r*_pmc_valu.csv:
    rocprofv3 --pmc SQ_INSTS_VALU_{ADD,MUL,FMA}_F64 / SQ_INSTS_VALU passes of this command at batch 256, tools/gpu_pmc.sh).
    Wave-level instruction counts; flops = 64 lanes x (ADD + MUL + 2 FMA)."""
    import csv
    path = newest_profile("r*_pmc_valu.csv")
    if not path or batch != 256:
        return None
    best = None
    with open(path) as f:
        for d in csv.DictReader(l for l in f if not l.startswith("#")):
            if d["kernel"].split("<")[0] == "ll::" + kernel and d.get("SQ_INSTS_VALU_FMA_F64_avg"):
                g = int(d["grid_threads"])
                if best is None or g > best[0]:
                    best = (g, d)
    if best is None:
        return None
    d = best[1]
    add, mul, fma = (float(d[f"SQ_INSTS_VALU_{k}_F64_avg"]) for k in ("ADD", "MUL", "FMA"))
    return {"fp64_wave_instructions": int(add + mul + fma), "valu_wave_instructions": int(float(d["SQ_INSTS_VALU_avg"])),
            "flops": 64.0 * (add + mul + 2.0 * fma), "source": os.path.relpath(path, ROOT)}


def pmc_knn_issue(batch: int):
    """The k-NN class is instruction-bound, not bandwidth-bound (its candidates are cache resident): VALU issue share and lane
    utilisation of the full-search kernel from the newest committed instruction counters (profiles/r*_pmc_valu.csv) and its
    average launch time from the newest committed kernel trace (profiles/r*_kernel_trace_by_grid.csv) -- both of this command at
    batch 256, both labelled.  None when either is missing."""
    import csv
    pv, pt = newest_profile("r*_pmc_valu.csv"), newest_profile("r*_kernel_trace_by_grid.csv")
    if not pv or not pt or batch != 256:
        return None

    def biggest(path, kernel):
        best = None
        with open(path) as f:
            for d in csv.DictReader(l for l in f if not l.startswith("#")):
                if d["kernel"].split("<")[0] == kernel and (best is None or int(d["grid_threads"]) > int(best["grid_threads"])):
                    best = d
        return best
    # round 4: the surface queries' search + the corner searches + the block build are ONE kernel (ll_knn_kernels.hip)
    c, t = biggest(pv, "ll::reg_knn_tile_kernel"), biggest(pt, "ll::reg_knn_tile_kernel")
    if not c or not t or not c.get("SQ_INSTS_VALU_avg") or not c.get("SQ_THREAD_CYCLES_VALU_avg") or not c.get("SQ_ACTIVE_INST_VALU_avg"):
        return None
    insts, us = float(c["SQ_INSTS_VALU_avg"]), float(t["avg_us"])
    return {"bound": "valu-issue", "kernel": "reg_knn_tile_kernel", "valu_wave_instructions_per_launch": int(insts),
            # one wave instruction occupies a SIMD for 4 cycles; 1024 SIMDs at 2.4 GHz
            "valu_issue_frac_at_2p4GHz": round(insts * 4.0 / 1024.0 / (us * 1e-6 * 2.4e9), 3),
            "lane_utilisation": round(float(c["SQ_THREAD_CYCLES_VALU_avg"]) / (64.0 * float(c["SQ_ACTIVE_INST_VALU_avg"])), 3),
            "avg_launch_us": us, "threads_per_launch": int(t["grid_threads"]),
            "sources": [os.path.relpath(pv, ROOT), os.path.relpath(pt, ROOT)]}


def pipeline_schedule(k_steps, n_slots, start, collect):
    """k_steps batches through n_slots slots (slot = extractor handle + registrar + voxel filters, each with its own streams):
    batch i + D - 1 is started (start(slot): extract, select, enqueue the registration) before batch i is collected (collect(slot):
    wait for it and download the results).  A slot is started again only after its previous batch has been collected; every batch
    is started once and collected once, in order.  Returns the last collect's value.  (tests/test_bench_helpers.py replays it.)"""
    D_, o = n_slots, None
    for j in range(min(D_ - 1, k_steps)):   # prologue: the first D - 1 batches
        start(j)
    for i in range(k_steps):
        if i + D_ - 1 < k_steps:
            start((i + D_ - 1) % D_)        # (the slot's previous batch, i - 1, was collected in the last iteration)
        o = collect(i % D_)
    return o


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def relaunch_under_torchrun(n, script=None):
    """`python bench.py --gpus N` without a launcher: start N ranks of this same command line under torch.distributed.run (one
    process per GPU, rendezvous on 127.0.0.1) and hand back their exit code.  Under a launcher (WORLD_SIZE set) nothing happens
    here: the ranks already exist."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(script or __file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def dry_run(args, rank, world):
    """Launcher / collective plumbing without the hot path (the CPU test of the --gpus path: gloo, no GPU): every rank reports a
    made-up elapsed time, the line is assembled exactly as the real one is (MAX over ranks, per-rank rates gathered)."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    elapsed = 0.010 * (rank + 1)
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    per = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(per, torch.tensor([args.batch * args.steps / elapsed], dtype=torch.float64))
    # the HIP device every rank's handles would be created on: main() passes `dev` = LOCAL_RANK to every ll_*_create (tests/test_bench_helpers.py
    # checks the call sites)
    devs = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(devs, torch.tensor([int(os.environ.get("LOCAL_RANK", "0"))], dtype=torch.int64))
    if rank == 0:
        print(json.dumps({"metric": "scans_per_s", "dry_run": True, "n_gpus": dist.get_world_size(), "requested_gpus": args.gpus,
                          "value": round(args.batch * args.steps * world / float(t.item()), 2),
                          "per_rank_scans_per_s": [round(float(x.item()), 2) for x in per], "devices": [int(x.item()) for x in devs]}))
    dist.barrier()
    dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.dry_run:
        return dry_run(args, rank, world)
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
    n_distinct = max(1, min(args.distinct_scans, B))
    base = [synth.make_scan(world_model, 1000 * rank + k, n=N) for k in range(n_distinct)]
    rng = np.random.default_rng(4242 + rank)
    # page-locked host copies of the batch: the streamed figure uploads from here with asynchronous copies
    scans_t = torch.empty((B, N, 4), dtype=torch.float32).pin_memory()
    scans = scans_t.numpy()
    for b in range(B):
        scans[b] = base[b % n_distinct].xyzi
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
    times = np.full(B, 1.0)
    fe.upload(scans, times)
    def make_registrar():
        r_ = Point_cloud_registration(max_scans=B, max_features=N, device=dev)
        q_ = r_.params
        q_.icp_max_iterations, q_.ceres_max_iterations, q_.force_all_iterations = args.icp_iters, 20, 1
        q_.para_max_angular_rate, q_.para_max_speed, q_.max_final_cost = 20.0, 0.3, 1000.0
        q_.current_frame_index, q_.mapping_init_accumulate_frames = 100, 50
        q_.maximum_allow_residual_block = N
        r_.set_profiling(True)
        if args.force_general or args.no_knn_coop or args.no_knn_tile or args.knn_tile_with_reuse:
            r_.set_debug(False, force_general_solver=args.force_general,
                         no_knn_coop=args.no_knn_coop, no_knn_tile=args.no_knn_tile, knn_tile_with_reuse=args.knn_tile_with_reuse)
        return r_

    reg = make_registrar()
    p = reg.params
    make_vox = lambda: (VoxelGrid(N, B, device=dev), VoxelGrid(N, B, device=dev)) if args.q_pipe else None
    vox = make_vox()

    def fe_stage(fe_, upload=False):
        if upload:
            fe_.upload(scans, times, wait=False)  # asynchronous copy from page-locked memory on the handle's stream, ahead of its kernels
        fe_.extract_batch(B)
        fe_.resolve()
        fe_.select_batch(B, -1, 0.0, 1.0)

    def enqueue(reg_, fe_, vox_):
        if vox_:
            reg_.enqueue_fe_downsampled(mp, fe_, vox_[0], vox_[1], 0.1, 0.4, B, init, init)
        else:
            reg_.enqueue_fe(mp, fe_, B, init, init)

    def run_on(fe_):
        fe_stage(fe_)
        enqueue(reg, fe_, vox)
        return reg.collect(B)

    def pipelined(k_steps, slots_, upload=False):
        return pipeline_schedule(k_steps, len(slots_), lambda j: (fe_stage(slots_[j][0], upload), enqueue(slots_[j][1], slots_[j][0], slots_[j][2])),
                                 lambda j: slots_[j][1].collect(B))

    def step():
        return run_on(fe)

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
    per_rank = [round(B * args.steps / elapsed, 2)]
    if dist is not None:
        mine = torch.tensor([B * args.steps / elapsed], device="cuda", dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [round(float(x.item()), 2) for x in allr]
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        world = dist.get_world_size()  # what RCCL saw
    res, pc, pi, reps = out
    nc, ns, nf, n_amb = fe.counts(B)
    nc_fe, ns_fe = nc, ns
    if vox:  # what the registrar saw
        nc, ns = vox[0].counts(B)[0], vox[1].counts(B)[0]
    total_scans = B * args.steps * world
    value = total_scans / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    sequential, pipeline_note = None, None
    seq_ms_per_step = ms_per_step

    # ---- the same K steps, consecutive batches overlapped: two extractor handles (both hold the resident scans) and two registrars
    #      on their own streams; batch i+1 is extracted and its registration enqueued while batch i's kernels run, so the device
    #      does not idle during the host's round trips (label fix-up after extraction, result download) and the tail of one
    #      batch's launches (a few slow scans on a few CUs) is filled by the other's.  Nothing is skipped: every batch's
    #      extraction, selection, 10 ICP iterations and result download happen inside the timed region, the first batch's
    #      extraction included.  The sequential figure above stays in the line as `sequential`; kernel times / roofline are
    #      taken from it (launches of two batches sharing the device stretch each other's event intervals).
    fe_b, slots = None, None
    if not args.no_pipeline:
        fe_b = Livox_laser(max_points=N, max_scans=B, device=dev, piecewise_number=1)
        fe_b.upload(scans, times)
        D = max(2, args.in_flight, args.q_pipe_in_flight if args.q_pipe else 0)
        slots = [(fe, reg, vox), (fe_b, make_registrar(), make_vox())]
        for _ in range(D - 2):
            f_ = Livox_laser(max_points=N, max_scans=B, device=dev, piecewise_number=1)
            f_.upload(scans, times)
            slots.append((f_, make_registrar(), make_vox()))

        pipelined(max(D, args.warmup), slots)
        barrier()
        tp = time.perf_counter()
        out_p = pipelined(args.steps, slots)
        barrier()
        el_p = time.perf_counter() - tp
        per_rank_p = [round(B * args.steps / el_p, 2)]
        if dist is not None:
            mine = torch.tensor([B * args.steps / el_p], device="cuda", dtype=torch.float64)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank_p = [round(float(x.item()), 2) for x in allr]
            t = torch.tensor([el_p], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el_p = float(t.item())
        same = bool(np.array_equal(out_p[0], res) and np.array_equal(out_p[1], pc) and np.array_equal(out_p[2], pi))
        sequential = {"value": round(value, 2), "unit": "scans/s", "ms_per_step": round(ms_per_step, 3), "per_rank_scans_per_s": per_rank,
                      "note": "one batch at a time: extract, fix up labels, select, register, download, then the next batch"}
        value, ms_per_step, per_rank = total_scans / el_p, 1e3 * el_p / args.steps, per_rank_p
        pipeline_note = {"batches_in_flight": D, "results_equal_sequential_bitwise": same,
                         "what": f"{D} extractor handles + {D} registrars on their own streams; batch i+1 is extracted and enqueued while batch i runs"}

    # ---- the same steps with the scans crossing PCIe inside the timed region (SURVEY 8d "end-to-end ... of one scan") ----
    streamed = None
    if not args.no_streamed:
        if slots is not None:
            # the same loop as the resident figure, every batch's 98 MB copied from page-locked host memory inside the timed region
            pipelined(len(slots), slots, upload=True)
            barrier()
            ts = time.perf_counter()
            pipelined(args.steps, slots, upload=True)
            barrier()
            el_s = time.perf_counter() - ts
            note_s = (f"every batch uploaded from page-locked host memory inside the timed region (asynchronous copy on the extractor handle's stream), "
                      f"{len(slots)} batches in flight: batch i+1 crosses PCIe and is extracted while batch i registers")
            t_sync = t_up = float("nan")
        else:
            fe2 = Livox_laser(max_points=N, max_scans=B, device=dev, piecewise_number=1)
            pair = (fe, fe2)
            for f_ in pair:  # warm-up of both handles' paths
                f_.upload(scans, times, wait=False)
                f_.sync()
                run_on(f_)
            barrier()
            ts = time.perf_counter()
            pair[0].upload(scans, times, wait=False)  # the first batch's copy is inside the timed region too
            t_sync = t_up = 0.0
            for i in range(args.steps):
                cur, nxt = pair[i % 2], pair[(i + 1) % 2]
                ta = time.perf_counter()
                cur.sync()                                # batch i has arrived
                tb = time.perf_counter()
                if i + 1 < args.steps:
                    nxt.upload(scans, times, wait=False)  # batch i+1 crosses PCIe while batch i is processed
                tc = time.perf_counter()
                t_sync += tb - ta
                t_up += tc - tb
                run_on(cur)
            barrier()
            el_s = time.perf_counter() - ts
            note_s = ("every batch uploaded from page-locked host memory inside the timed region (asynchronous copies, two extractor "
                      "handles: batch i+1 crosses PCIe while batch i runs; one registration at a time)")
        if dist is not None:
            t = torch.tensor([el_s], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el_s = float(t.item())
        streamed = {"value": round(total_scans / el_s, 2), "unit": "scans/s", "ms_per_step": round(1e3 * el_s / args.steps, 3),
                    "h2d_bytes_per_step": int(B * N * 16), "note": note_s}
        if t_sync == t_sync:
            streamed.update(host_ms_per_step_waiting_for_the_upload=round(1e3 * t_sync / args.steps, 3),
                            host_ms_per_step_in_the_upload_call=round(1e3 * t_up / args.steps, 3))
        fe.upload(scans, times)  # leave the first handle as the resident configuration left it

    # single-scan latency (batch of 1 through the same code path)
    lat = []
    fe1 = Livox_laser(max_points=N, max_scans=1, device=dev, piecewise_number=1)
    fe1.upload(scans[:1], np.full(1, 1.0))
    reg1 = Point_cloud_registration(max_scans=1, max_features=N, device=dev)
    for f_ in ("icp_max_iterations", "ceres_max_iterations", "force_all_iterations", "para_max_angular_rate", "para_max_speed",
               "max_final_cost", "current_frame_index", "mapping_init_accumulate_frames", "maximum_allow_residual_block"):
        setattr(reg1.params, f_, getattr(p, f_))
    if args.no_solver_groups or args.no_knn_coop or args.no_knn_tile or args.knn_tile_with_reuse:
        reg1.set_debug(False, no_solver_groups=args.no_solver_groups, no_knn_coop=args.no_knn_coop, no_knn_tile=args.no_knn_tile,
                       knn_tile_with_reuse=args.knn_tile_with_reuse)
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

    # secondary figure: the same workload in the other query mode (SURVEY 8d reports both) -- the reference's operating point: features
    # voxel-filtered on the device before registration (laser_mapping.hpp:1367-1373), a few hundred residual blocks per scan.  Such scans
    # take the small-scan solver (ll_reg_small_kernels.hip): one or two wavefronts per scan, several scans per CU -- which needs more scans
    # than CUs to pay, so the figure is taken at --q-pipe-batch scans per batch (default 2048) as well as at the headline's batch size.
    q_pipe_extra = None
    if not vox and not args.no_q_pipe:
        def q_pipe_figure(Bq, shipped_cap=False, in_flight=0, audit=False):
            rq = np.random.default_rng(777 + rank)
            idx = np.arange(Bq) % B
            init_q = init if Bq == B else np.stack([
                synth.pose_compose(poses_true[i], np.r_[synth.quat_from_axis_angle(rq.normal(size=3), np.deg2rad(rq.uniform(0, 1.0))), rq.uniform(-0.1, 0.1, 3)])
                for i in idx])

            def make_slot_q():
                f_ = Livox_laser(max_points=N, max_scans=Bq, device=dev, piecewise_number=1)
                for b0 in range(0, Bq, B):  # (the batch's scans repeat the headline's distinct ones; every slot has its own initial guess)
                    f_.upload(scans[:min(B, Bq - b0)], times[:min(B, Bq - b0)], first_scan=b0)
                r_ = Point_cloud_registration(max_scans=Bq, max_features=N, device=dev)
                for f in ("icp_max_iterations", "ceres_max_iterations", "force_all_iterations", "para_max_angular_rate", "para_max_speed",
                          "max_final_cost", "current_frame_index", "mapping_init_accumulate_frames", "maximum_allow_residual_block"):
                    setattr(r_.params, f, getattr(p, f))
                if shipped_cap:  # config/performance_precision.yaml:23 -- what cpu_baseline_shipped_config[_allcores] run
                    r_.params.maximum_allow_residual_block, r_.params.subsample_seed = 200, 7
                r_.set_profiling(True)
                return f_, r_, (VoxelGrid(N, Bq, device=dev), VoxelGrid(N, Bq, device=dev))

            def start_q(sl):
                sl[0].extract_batch(Bq); sl[0].resolve(); sl[0].select_batch(Bq, -1, 0.0, 1.0)
                sl[1].enqueue_fe_downsampled(mp, sl[0], sl[2][0], sl[2][1], 0.1, 0.4, Bq, init_q, init_q)

            sl0 = make_slot_q()
            start_q(sl0); sl0[1].collect(Bq)
            barrier()
            tq = time.perf_counter()
            kq_ms, kq_n = np.zeros(3), np.zeros(3)
            for _ in range(3):
                start_q(sl0)
                out_q = sl0[1].collect(Bq)
                kq_ms += sl0[1].kernel_times()[0]
                kq_n += sl0[1].kernel_times()[1]
            barrier()
            tq = (time.perf_counter() - tq) / 3
            ncq, nsq = sl0[2][0].counts(Bq)[0], sl0[2][1].counts(Bq)[0]
            # the small-scan solver against the HBM roofline, SURVEY 8(d)'s accounting: every residual block's constants once per launch
            # (65 B per line block, 48 B per plane block) + 224 B of state out per scan; duration = HIP events around the solver launches
            solver_ms = float(kq_ms[1] / max(1.0, kq_n[1]))
            alg_q = float(np.sum([65.0 * r.corner_avail + 48.0 * r.surf_avail + 224.0 for r in out_q[3]]))
            fig = {"batch": Bq, "scans_per_s_this_rank": round(Bq / tq, 1), "ms_per_step": round(1e3 * tq, 3),
                   "features_per_scan": {"corner": float(ncq.mean()), "surface": float(nsq.mean())},
                   "blocks_per_scan_last_iteration": float(np.mean([r.n_blocks_last for r in out_q[3]])),
                   "accepted_frac": float(np.mean(out_q[0])), "lm_iters_per_scan": float(np.mean([r.lm_iterations_total for r in out_q[3]])),
                   "kernel_ms_per_step": {"knn+build": round(float(kq_ms[0] / 3), 3), "solver": round(float(kq_ms[1] / 3), 3)},
                   "solver_roofline": {"bound": "hbm", "kernel": "reg_solve_small_kernel", "avg_launch_ms": round(solver_ms, 4),
                                       "algorithmic_bytes_per_launch": int(alg_q), "achieved": round(alg_q / (solver_ms * 1e-3) / 1e9, 2), "unit": "GB/s",
                                       "peak": 8000.0, "frac": round(alg_q / (solver_ms * 1e-3) / 1e9 / 8000.0, 5),
                                       "limited_by": "dependent fp64 issue of the per-scan Levenberg-Marquardt controller and the evaluations of "
                                                     "a few hundred blocks per wavefront (latency chains), not HBM bandwidth"}}
            if audit:
                fig["parity_audit_vs_oracle"] = qpipe_oracle_audit(args, synth, corner, surf, scans, idx, init_q, out_q, p)
            if in_flight > 1:
                # A voxel-filtered batch ends with its slowest scan's last line search; batches are independent: with several in flight the
                # CUs of the early finishers run the other batches' scans
                slots_q = [sl0] + [make_slot_q() for _ in range(in_flight - 1)]
                pipeline_schedule(in_flight, in_flight, lambda j: start_q(slots_q[j]), lambda j: slots_q[j][1].collect(Bq))
                barrier()
                tq = time.perf_counter()
                out_p = pipeline_schedule(3 * in_flight, in_flight, lambda j: start_q(slots_q[j]), lambda j: slots_q[j][1].collect(Bq))
                barrier()
                tq = (time.perf_counter() - tq) / (3 * in_flight)
                fig = {**fig, "one_batch_at_a_time": {k_: fig[k_] for k_ in ("scans_per_s_this_rank", "ms_per_step")},
                       "scans_per_s_this_rank": round(Bq / tq, 1), "ms_per_step": round(1e3 * tq, 3), "batches_in_flight": in_flight,
                       "results_equal_one_at_a_time_bitwise": bool(np.array_equal(out_p[1], out_q[1]))}
                for sl in slots_q[1:]:
                    sl[0].close(); sl[1].close(); sl[2][0].close(); sl[2][1].close()
            sl0[0].close(); sl0[1].close(); sl0[2][0].close(); sl0[2][1].close()
            return fig

        Bq = max(B, args.q_pipe_batch)
        q_pipe_extra = q_pipe_figure(Bq, in_flight=(max(2, args.q_pipe_in_flight) if slots is not None else 0),
                                     audit=(rank == 0 and world == 1 and not args.no_cpu_baseline))
        q_pipe_extra["note"] = ("device VoxelGrid (leaf 0.1 / 0.4 m, laser_mapping.hpp:1367-1373) between extraction and registration; small-scan solver "
                                "(one / two wavefronts per scan in batches of >= 512 scans, four below)")
        if Bq != B:
            # (a batch of 256 small scans is one scan per CU and ends with its slowest scan: only batches in flight fill the device)
            q_pipe_extra["at_the_headline_batch_size"] = q_pipe_figure(B, in_flight=(8 if slots is not None else 0))
        # like-for-like with cpu_baseline_shipped_config[_allcores]: the same features capped at maximum_residual_blocks = 200
        q_pipe_extra["shipped_config_200_blocks"] = q_pipe_figure(Bq, shipped_cap=True)

    # ---- roofline of the dominant kernel (HIP events on the registrar's stream, see ll_reg_set_profiling) ----
    names = ["reg_knn_build_kernel", "reg_solve_kernel", "reg_finalize_kernel"]
    # the k-NN class is several kernels (transform / search / re-query / block build); the single kernel with the largest
    # share of the step is the solver as long as its time exceeds the largest k-NN kernel's (profiles/*_kernel_trace_by_grid.csv)
    dom = 1 if k_ms[1] >= 0.45 * k_ms[0] else 0
    avg_ms = float(k_ms[dom] / max(1.0, k_n[dom]))
    line_blocks = float(sum(r.corner_avail for r in reps))
    plane_blocks = float(sum(r.surf_avail for r in reps))
    queries = float(nc.sum() + ns.sum())
    compact = not args.force_general
    # residual-block constants as stored (data layout): plane blocks 48 B packed (fp32 point, fp64 normal and
    # offset; round-1 layout: the same 48 B in three planes + 1 B flag), line blocks 65 B (16 B point + 1 B flag + 48 B a', u')
    plane_bytes = 48.0 if compact else 49.0
    block_bytes = line_blocks * 65.0 + plane_blocks * plane_bytes
    if dom == 1:
        # algorithmic bytes of one solver launch: every residual block's constants read once + the 28 reduced doubles
        # per scan (224 B per scan and iteration)
        alg_bytes = block_bytes + B * 224.0
    else:
        # k-NN + block build launch: 16 B/query in, the block constants out (candidate gather is cache-resident)
        alg_bytes = queries * 16.0 + block_bytes
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic_bytes(names[dom], B)
    roofline = {"bound": "hbm", "kernel": names[dom], "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 6), "traffic": traffic,
                "traffic_source": traffic_src,  # a committed rocprofv3 PMC summary of this command, not measured in this run
                "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(alg_bytes),
                "plane_block_bytes": plane_bytes,
                # the contract's roofline is bytes against the HBM peak; what the counters say limits this kernel (profiles/):
                "limited_by": ("fp64 VALU issue of the cost evaluations + per-launch table build + serial phases, not HBM bandwidth"
                               if dom == 1 else "VALU issue of the searches (map patch is cache resident), not HBM bandwidth"),
                # what the kernel really moves (rocprofv3 PMC passes, profiles/) against the same peak
                "traffic_frac": None if traffic is None else round(traffic / (avg_ms * 1e-3) / 8e12, 4)}
    roofline.update(traffic_stamp(traffic_src))  # does the committed counter summary belong to the kernels running now?

    # The solver's arithmetic is fp64 VALU (no MFMA shape on this path): the same launch against the fp64 vector peak
    # (256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz = 78.6 TFLOP/s = half the guide's 157.3 TFLOP/s fp32 vector figure).
    vf = pmc_valu_fp64(names[dom], B) if dom == 1 else None
    if vf is not None:
        tf = vf["flops"] / (avg_ms * 1e-3) / 1e12
        # share of the launch during which a SIMD issues VALU work at all: one wave instruction = 4 cycles, 1024 SIMDs
        busy = vf["valu_wave_instructions"] * 4.0 / 1024.0 / (avg_ms * 1e-3 * 2.4e9)
        roofline["valu_fp64"] = {"achieved": round(tf, 2), "peak": 78.6, "unit": "TFLOP/s", "frac": round(tf / 78.6, 4),
                                 "fp64_wave_instructions_per_launch": vf["fp64_wave_instructions"],
                                 "valu_issue_busy_frac_at_2p4GHz": round(busy, 3), "source": vf["source"]}

    # ---- whole-path algorithmic bytes per scan, U and C counted per scan ----
    roofline_path = None
    try:
        from oracle import orc_roofline
        cell_c = args.cell_corner if args.cell_corner > 0 else 1.45
        cell_s = args.cell_surf if args.cell_surf > 0 else 0.6
        mc, ms_ = orc_roofline.MapCells(corner, cell_c), orc_roofline.MapCells(surf, cell_s)
        n_uc = min(B, 32)
        UC = [orc_roofline.scan_u_c(mc, ms_, scans[b], init[b]) for b in range(n_uc)]
        U = float(np.mean([u for u, c, q in UC]))
        Cc = float(np.mean([c for u, c, q in UC]))
        Qs = float(np.mean([q for u, c, q in UC]))
        b_iter = 16.0 * Qs + 12.0 * U + 8.0 * Cc + 40.0 * Qs + 224.0
        b_scan = 16.0 * N + 8.0 * N + 4.0 * Qs + args.icp_iters * b_iter
        roofline_path = {"definition": "SURVEY 8(d): B_scan = 16N + 8N + 4(nC+nS) + iters * (16Q + 12U + 8C + 40Q + 224); achieved = B_scan * scans/s",
                         "U_distinct_map_points_in_27_cell_neighbourhoods": round(U, 1), "C_distinct_cells": round(Cc, 1), "Q_queries": round(Qs, 1),
                         "scans_counted": n_uc, "cells": {"corner_m": cell_c, "surface_m": cell_s},
                         "B_iter_bytes": int(b_iter), "B_scan_bytes": int(b_scan), "achieved": round(b_scan * value / 1e9, 3), "unit": "GB/s",
                         "peak": 8000.0, "frac": round(b_scan * value / 8e12, 6)}
    except Exception as e:  # the figure is reporting, never a reason to lose the line
        roofline_path = {"error": repr(e)}

    result = {
        "metric": "scans_per_s", "value": round(value, 2), "unit": "scans/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 points / f64 solve", "data": "synthetic",
        "config": {"workload": "C2: 24k-pt Mid-40 scan vs 5M-pt corner+surface map, 10 ICP iters (fixed), k=5, "
                               + ("Q-pipe (device VoxelGrid 0.1/0.4 before registration)" if vox else "Q-full"),
                   "scan_points": N, "map_points": int(len(corner) + len(surf)), "map_corner": int(len(corner)),
                   "map_surf": int(len(surf)), "icp_iters": args.icp_iters, "batch_scans_per_step_per_gpu": B,
                   "distinct_scans_in_batch": n_distinct,
                   "parallelism": f"replicas x{world} (independent scans, no data-path collective)"
                                  + ("" if pipeline_note is None else f"; {pipeline_note['batches_in_flight']} batches in flight per GPU (extraction / registration of batch i+1 overlaps registration of batch i)")},
        "per_rank_scans_per_s": per_rank,
        # a device-side clock beside the wall clock: HIP events on the registrar's stream around every launch class, summed (the
        # extraction kernels run on the extractor's stream and are not in it: ~0.3 ms per step)
        "device_ms_per_step_hip_events": {"registrar_kernels": round(float(k_ms.sum() / args.steps), 3), "wall_one_batch_at_a_time": round(seq_ms_per_step, 3)},
        "sequential": sequential,
        "pipeline": pipeline_note,
        "roofline": roofline,
        "roofline_path": roofline_path,
        "roofline_knn": pmc_knn_issue(B),  # the other half of the step: instruction-bound (committed counters, labelled)
        "streamed": streamed,
        "q_pipe": q_pipe_extra,
        "kernel_ms_per_step": {names[i]: round(float(k_ms[i] / args.steps), 3) for i in range(3)},
        "per_iter_knn_jtj_ms_per_batch": round(float((k_ms[0] + k_ms[1]) / args.steps / max(1, args.icp_iters)), 4),
        "per_iter_ms_split_per_batch": {"transform_knn_build": round(float(k_ms[0] / args.steps / max(1, args.icp_iters)), 4),
                                        "solve": round(float(k_ms[1] / args.steps / max(1, args.icp_iters)), 4)},
        "single_scan_latency_ms": round(latency_ms, 3),
        "single_scan_solver": ("small-scan solver, four wavefronts (<= 1024 features)" if vox else
                               ("one workgroup per scan" if args.no_solver_groups else "group of 8 workgroups per scan (batches <= 16)")),
        "features_per_scan": {"corner": float(nc.mean()), "surface": float(ns.mean())},
        "knn_reuse_last_iter": dict(zip(("searched", "resorted"), reg.debug_worklists(B)), queries=int(nc.sum() + ns.sum()),
                                    corner_searched_resorted=reg.debug_worklists_by_kind(B)[0], corner_queries=int(nc.sum())),
        "accepted_frac": float(np.mean(res)), "lm_iters_per_scan": float(np.mean([r.lm_iterations_total for r in reps])),
        "ambiguous_labels": int(n_amb), "setup_s": {"synthetic_data": round(t_data, 1), "map_upload_grid_build": round(t_map, 2)},
    }

    cyc1, cycB = [int(v) for v in reg1.debug_cycles(0)], [int(v) for v in reg.debug_cycles(0)]
    if any(cyc1) or any(cycB):  # only the -DLL_SOLVE_TIMING build (LOAM_LIVOX_LIB=...timing.so) fills these
        result["single_scan_solver_phase_cycles"] = cyc1
        result["solver_phase_cycles_scan0"] = cycB
        tot = np.array([int(reg.debug_cycles(b)[5]) for b in range(B)], np.float64)  # slot 5: the whole solver call, summed over the launches of a registration
        ctl = np.array([int(reg.debug_cycles(b)[1]) for b in range(B)], np.float64)
        allc = np.array([[int(v) for v in reg.debug_cycles(b)] for b in range(B)], np.float64)
        result["solver_phase_cycles_mean_over_scans"] = [int(v) for v in allc.mean(0)]
        result["solver_phase_cycles_of_the_slowest_scan"] = [int(v) for v in allc[int(np.argmax(allc[:, 5]))]]
        qs = [0, 10, 25, 50, 75, 90, 99, 100]
        result["phase_cycles_slot14_quantiles_over_scans"] = [int(v) for v in np.percentile(allc[:, 14], qs)]  # (-DLL_TILE_TIMING: listed lanes per registration)
        result["solver_cycles_per_registration_quantiles"] = {"q": qs, "total": [int(v) for v in np.percentile(tot, qs)], "mean_total": int(tot.mean()),
                                                              "lm_controller": [int(v) for v in np.percentile(ctl, qs)], "mean_lm_controller": int(ctl.mean())}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # the CPU leg belongs to the N = 1 line only
        result.update(cpu_legs(args, synth, corner, surf, scans, init, B, vox, pc, res, reps, nc, ns, nc_fe, ns_fe))
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


_ORACLE_TREES = {}


def oracle_trees(corner, surf):
    """the oracle's k-d trees of the bench map, built once per process (the Q-pipe audit and the CPU legs share them)"""
    from oracle import orc
    if "t" not in _ORACLE_TREES:
        tb = time.perf_counter()
        _ORACLE_TREES["t"] = (orc.KdTree(corner), orc.KdTree(surf), 0.0)
        _ORACLE_TREES["t"] = _ORACLE_TREES["t"][:2] + (time.perf_counter() - tb,)
    return _ORACLE_TREES["t"]


def qpipe_oracle_audit(args, synth, corner, surf, scans, idx, init_q, out_q, p):
    """Every scan of a Q-pipe batch (voxel-filtered features, small-scan solver) against the CPU oracle run from the same raw scan and the same
    initial guess: pose, accept / reject, residual-block and iteration counts (VERDICT r5 next #9 ii: the Q-pipe legs printed bit-equality
    between their own runs but no oracle audit).  One oracle run per slot, spread over the host's cores (ctypes releases the GIL)."""
    from oracle import orc
    tc, ts, _ = oracle_trees(corner, surf)
    res_q, pc_q, _, reps_q = out_q
    Bq = len(idx)
    prm = orc.RegParams.defaults(icp_iters=int(p.icp_max_iterations), ceres_iters=int(p.ceres_max_iterations), force_all=int(p.force_all_iterations))
    prm.max_final_cost = float(p.max_final_cost)
    feats = {}
    lock = threading.Lock()

    def features(i):
        with lock:
            hit = feats.get(i)
        if hit is None:
            o = orc.fe_extract(scans[i], 1.0)
            ci, si, _ = orc.fe_get_features(o, 0.0, 1.0)
            hit = (orc.voxel_grid(orc.feature_cloud(o, ci), 0.1)[1], orc.voxel_grid(orc.feature_cloud(o, si), 0.4)[1])
            with lock:
                feats[i] = hit
        return hit

    rows = [None] * Bq
    n_thr = max(1, min(Bq, args.cpu_threads if args.cpu_threads > 0 else (os.cpu_count() or 1)))

    def worker(t):
        for b in range(t, Bq, n_thr):
            fc_o, fs_o = features(int(idx[b]))
            ret, opc, _, orep = orc.reg_solve(tc, ts, fc_o, fs_o, prm, init_q[b], init_q[b])
            dt, dr = synth.pose_error(pc_q[b], opc)
            rows[b] = (dt, dr, ret == res_q[b], orep.n_blocks_last == reps_q[b].n_blocks_last and orep.corner_avail == reps_q[b].corner_avail
                       and orep.surf_avail == reps_q[b].surf_avail, orep.lm_iterations_total == reps_q[b].lm_iterations_total,
                       orep.icp_iterations == reps_q[b].icp_iterations)

    tb = time.perf_counter()
    th = [threading.Thread(target=worker, args=(t,)) for t in range(n_thr)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    r = np.array([[float(v) for v in row] for row in rows])
    worst = np.argsort(-r[:, 0])[:4]
    return {"scans": Bq, "max_pose_err_m": float(r[:, 0].max()), "max_pose_err_rad": float(r[:, 1].max()),
            "pose_err_m_quantiles": {"q": [50, 90, 99, 99.9], "v": [float(v) for v in np.percentile(r[:, 0], [50, 90, 99, 99.9])]},
            "scans_beyond_1e-7_m": int((r[:, 0] > 1e-7).sum()), "scans_beyond_1e-4_m": int((r[:, 0] > 1e-4).sum()),
            "worst_scans": [{"slot": int(b), "distinct_scan": int(idx[b]), "pose_err_m": float(r[b, 0]), "lm_equal": bool(r[b, 4]), "icp_equal": bool(r[b, 5]),
                             "device_lm": int(reps_q[b].lm_iterations_total), "device_final_cost": float(reps_q[b].final_cost), "accepted": int(res_q[b])} for b in worst],
            "accept_reject_identical": bool(r[:, 2].all()), "block_counts_identical": bool(r[:, 3].all()),
            "lm_iteration_counts_identical": int(r[:, 4].sum()), "icp_iteration_counts_identical": int(r[:, 5].sum()),
            "host_threads": n_thr, "seconds": round(time.perf_counter() - tb, 1)}


def cpu_legs(args, synth, corner, surf, scans, init, B, vox, pc, res, reps, nc, ns, nc_fe, ns_fe):
    """CPU baseline: the oracle (C restatement of the reference algorithm, gcc -O3 -- pinned to the reference's own code,
    oracle/README.md; the reference binary itself needs PCL / Ceres / ROS) on the host cores of this box.  It is the
    checker, never the product."""
    from oracle import orc
    out = {}
    tc, ts, t_tree = oracle_trees(corner, surf)

    def one_scan(b, prm, q_pipe=False):
        o = orc.fe_extract(scans[b], 1.0)
        ci, si, fi = orc.fe_get_features(o, 0.0, 1.0)
        fc_o, fs_o = orc.feature_cloud(o, ci), orc.feature_cloud(o, si)
        if q_pipe:
            fc_o, fs_o = orc.voxel_grid(fc_o, 0.1)[1], orc.voxel_grid(fs_o, 0.4)[1]
        ret, opc, _, orep = orc.reg_solve(tc, ts, fc_o, fs_o, prm, init[b], init[b])
        return ret, opc, orep, len(ci), len(si), len(fc_o), len(fs_o)

    # -- (i) one thread: median of >= 20 runs after 3 warm-ups; the same runs are the parity audit against the
    #        reference arithmetic (exact fp64 plane normals)
    prm = orc.RegParams.defaults(icp_iters=args.icp_iters, ceres_iters=20, force_all=1)
    prm.max_final_cost = 1000.0
    for b in range(min(3, B)):
        one_scan(b, prm, bool(vox))
    t_runs, errs, same_sets, same_blocks, same_res, same_lm = [], [], True, True, True, True
    n_runs = max(1, args.cpu_runs)
    for k in range(n_runs):
        b = k % B
        tb = time.perf_counter()
        ret, opc, orep, n_ci, n_si, n_fc, n_fs = one_scan(b, prm, bool(vox))
        t_runs.append(time.perf_counter() - tb)
        errs.append(synth.pose_error(pc[b], opc))
        same_sets &= (n_ci == nc_fe[b] and n_si == ns_fe[b] and n_fc == nc[b] and n_fs == ns[b])
        same_blocks &= (orep.n_blocks_last == reps[b].n_blocks_last and orep.corner_avail == reps[b].corner_avail
                        and orep.surf_avail == reps[b].surf_avail)
        same_lm &= (orep.lm_iterations_total == reps[b].lm_iterations_total and orep.icp_iterations == reps[b].icp_iterations)
        same_res &= (ret == res[b])
    med = float(np.median(t_runs))
    out["cpu_baseline_1thread"] = {"value": round(1.0 / med, 4), "unit": "scans/s", "cores": 1, "kind": "port",
                                   "ms_per_scan_median": round(1e3 * med, 1), "runs": n_runs, "warmups": 3,
                                   "sample": f"median of {n_runs} single-scan runs (extract + {args.icp_iters} ICP iters, Q-full) of the step's scans "
                                             f"vs the same {len(corner) + len(surf)}-pt map; k-d tree build {t_tree:.1f}s excluded"}

    # -- (ii) all host cores, one scan per thread (the reference's maximum_parallel_thread model, laser_mapping.hpp:1737)
    n_thr = args.cpu_threads if args.cpu_threads > 0 else (os.cpu_count() or 1)
    per_thread = 2 if n_thr >= 16 else 4
    done = [0] * n_thr

    audit = [[] for _ in range(n_thr)]  # the same runs extend the parity audit to every scan they touch

    def worker(t):
        for j in range(per_thread):
            b = (t * per_thread + j) % B
            ret, opc, orep, n_ci, n_si, n_fc, n_fs = one_scan(b, prm, bool(vox))
            done[t] += 1
            audit[t].append((b, synth.pose_error(pc[b], opc),
                             n_ci == nc_fe[b] and n_si == ns_fe[b] and n_fc == nc[b] and n_fs == ns[b],
                             orep.n_blocks_last == reps[b].n_blocks_last and orep.corner_avail == reps[b].corner_avail and orep.surf_avail == reps[b].surf_avail,
                             orep.lm_iterations_total == reps[b].lm_iterations_total and orep.icp_iterations == reps[b].icp_iterations,
                             ret == res[b]))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_thr)]
    tb = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    t_all = time.perf_counter() - tb
    out["cpu_baseline"] = {"value": round(sum(done) / t_all, 4), "unit": "scans/s", "cores": n_thr, "kind": "port",
                           "sample": f"{sum(done)} scans of the step on {n_thr} threads, one scan per thread at a time ({per_thread} each), same map, "
                                     f"shared read-only k-d trees; {t_all:.1f} s wall",
                           "host_cores_available": os.cpu_count()}

    # -- (iii) the shipped operating point: Q-pipe features, maximum_residual_blocks = 200 (config/performance_*.yaml)
    prm_s = orc.RegParams.defaults(icp_iters=args.icp_iters, ceres_iters=20, force_all=1)
    prm_s.max_final_cost = 1000.0
    prm_s.maximum_allow_residual_block = 200
    prm_s.subsample_seed = 7
    for b in range(min(3, B)):
        one_scan(b, prm_s, True)
    ts_runs = []
    for k in range(n_runs):
        tb = time.perf_counter()
        one_scan(k % B, prm_s, True)
        ts_runs.append(time.perf_counter() - tb)
    med_s = float(np.median(ts_runs))
    out["cpu_baseline_shipped_config"] = {"value": round(1.0 / med_s, 3), "unit": "scans/s", "cores": 1, "kind": "port",
                                          "ms_per_scan_median": round(1e3 * med_s, 2),
                                          "sample": f"median of {n_runs} runs: voxel-filtered features (0.1 / 0.4 m) and maximum_residual_blocks = 200 "
                                                    "(the shipped configs), the reference's real operating point"}
    # -- (iv) the shipped operating point on all host cores, one scan per thread at a time (laser_mapping.hpp:1737): the honest
    #         neighbour of the device's Q-pipe figure
    per_thread_s = 16
    done_s = [0] * n_thr

    def worker_s(t):
        for j in range(per_thread_s):
            one_scan((t * per_thread_s + j) % B, prm_s, True)
            done_s[t] += 1

    threads = [threading.Thread(target=worker_s, args=(t,)) for t in range(n_thr)]
    tb = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    t_all_s = time.perf_counter() - tb
    out["cpu_baseline_shipped_config_allcores"] = {"value": round(sum(done_s) / t_all_s, 3), "unit": "scans/s", "cores": n_thr, "kind": "port",
                                                   "sample": f"{sum(done_s)} scans on {n_thr} threads ({per_thread_s} each, one scan per thread at a time): "
                                                             f"voxel-filtered features, maximum_residual_blocks = 200; {t_all_s:.1f} s wall",
                                                   "host_cores_available": os.cpu_count()}
    seen = set(range(min(n_runs, B)))
    for lst in audit:
        for b, e, s_sets, s_blocks, s_lm, s_res in lst:
            seen.add(b)
            errs.append(e)
            same_sets &= bool(s_sets)
            same_blocks &= bool(s_blocks)
            same_lm &= bool(s_lm)
            same_res &= bool(s_res)
    out["parity_vs_cpu"] = {"max_pose_err_m": float(max(e[0] for e in errs)), "max_pose_err_rad": float(max(e[1] for e in errs)),
                            "feature_counts_identical": bool(same_sets), "lm_and_icp_iteration_counts_identical": bool(same_lm),
                            "residual_block_counts_identical": bool(same_blocks), "accept_reject_identical": bool(same_res),
                            "scans_compared": len(seen),
                            "note": "device results of the timed step against the oracle: the single-thread runs plus every scan of the all-core leg"}
    return out


if __name__ == "__main__":
    main()
