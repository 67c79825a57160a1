#!/usr/bin/env python3
"""BASELINE config C4's unit on MI355X: independent sequences with LOCAL MAP GROWTH, one sequence per GPU.

Every rank runs loam_livox_amd.mapping.Laser_mapping (extract -> device VoxelGrid -> register -> history add ->
match-buffer refresh, laser_mapping.hpp:1311-1520 / 460-566, all resident in HBM) over its own synthetic sequence;
there is no data-path collective.  The one exchange step is the gather of the ranks' CELL MAPS at the end, as BASELINE config C4
words it: m_pt_cell_map_corners / m_pt_cell_map_planes, which receive every registered frame (laser_mapping.hpp:1492-1493) and so
hold the map the whole sequence built (the 20-frame history match buffer is only its newest slice).  They are read where they lie on
the device (ll_cellmap_device_view: points in (cell, insertion) order + the 64-bit cell key of every point) and exchanged by
loam_livox_amd.multigpu.gather_cell_maps -- counts all-gather + grouped point-to-point sends, RCCL when --gpus > 1, no host hop --
which returns the union in cell-map layout.  --no-cell-maps: the round-4 form (no cell maps kept; the match buffer is gathered).

  python bench_c4.py [--frames F]                                  (1 GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench_c4.py --gpus N

Prints one JSON line: frames/s over all ranks (a frame = one 24k-point scan through the whole loop, strictly
sequential within a sequence), ms per frame, drift against the synthetic ground truth, sub-map sizes."""
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


def pose_inv(p, synth):
    R = synth.quat_to_mat(p[:4])
    return np.r_[-p[0], -p[1], -p[2], p[3], -(R.T @ p[4:])]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--scan-points", type=int, default=24000)
    ap.add_argument("--history", type=int, default=20)
    ap.add_argument("--plane-res", type=float, default=0.15)
    ap.add_argument("--line-res", type=float, default=0.1)
    ap.add_argument("--matching-mode", type=int, default=0, help="0 = history match buffer (shipped configs), 1 = cell maps (laser_mapping.hpp:689)")
    ap.add_argument("--max-blocks", type=int, default=0, help="optimization/maximum_residual_blocks (200 in the shipped configs): random "
                    "sub-sampling of the features on the library's reproducible stream; 0 = every feature is a residual block")
    ap.add_argument("--distinct-frames", type=int, default=0, help="generate only this many distinct scans (a multiple of 50, the period of the "
                    "out-and-back trajectory) and replay them: frame k uses scan k mod D, whose pose is frame k's; 0 = every frame its own scan")
    ap.add_argument("--no-prefetch", action="store_true", help="A/B: extract frame k + 1 only after frame k has been registered")
    ap.add_argument("--no-cell-maps", action="store_true", help="A/B: do not maintain the corner / surface cell maps (matching mode 0 never reads them); "
                    "the gather then exchanges the history match buffer")
    ap.add_argument("--cpu-frames", type=int, default=8, help="frames of rank 0's sequence also run through the CPU oracle (0 = skip)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # no launcher: start the ranks ourselves (bench.py's helper), one per GPU
        from bench import relaunch_under_torchrun
        sys.exit(relaunch_under_torchrun(args.gpus, __file__))
    import torch
    from loam_livox_amd import synth
    from loam_livox_amd.mapping import Laser_mapping
    from loam_livox_amd.multigpu import gather_cell_maps, gather_submaps
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    N, F = args.scan_points, args.frames

    world_model = synth.world_for_map_size(200_000)
    rng = np.random.default_rng(9000 + rank)
    start = synth.sensor_pose_in_world(world_model, rng)
    sgn = 1.0 if rank % 2 == 0 else -1.0
    step = np.r_[synth.quat_from_axis_angle(np.array([0.1, 0.2, 1.0]), np.deg2rad(0.2 * sgn)), np.array([-0.02, 0.01 * sgn, 0.0])]  # backing away: the view widens
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)
    scans, truth, cur = [], [], start
    step_back = pose_inv(step, synth)
    D = args.distinct_frames if args.distinct_frames > 0 else F
    assert D >= F or (D % 50 == 0 and D >= 100), "--distinct-frames must be a multiple of 50 and at least 100"
    for k in range(F):
        if k >= 3:  # 25 frames out, 25 frames back: a long sequence stays inside the synthetic room
            cur = synth.pose_compose(cur, step if ((k - 3) // 25) % 2 == 0 else step_back)
        # the trajectory has period 50 from frame 3 on: frame k >= D + 3 is taken from the same pose as frame k - D
        scans.append(scans[k - D] if k >= D + 3 else
                     synth.make_moving_scan(world_model, 7000 + 1000 * rank + k, N, inc_true=ident, pose_start=cur, t_phase=0.13 * k).xyzi)
        truth.append(synth.pose_compose(pose_inv(start, synth), cur))

    args_map = dict(maximum_history_size=args.history, init_accumulate_frames=2, line_res=args.line_res, plane_res=args.plane_res,
                    icp_max_iterations=10, ceres_max_iterations=20, max_allow_incre_R=20.0, max_allow_incre_T=0.3,
                    minimum_icp_R_diff=1e-3, minimum_icp_T_diff=1e-4, matching_mode=args.matching_mode,
                    maximum_in_fov_angle=45.0, maximum_residual_blocks=args.max_blocks, keep_cell_maps=not args.no_cell_maps)  # the ICP-diff defaults (0.01 deg / 1 cm, PCR:94-95) stop the ICP a
    # centimetre short of convergence every frame, and the lag accumulates in a map grown from those poses
    lm = Laser_mapping(scan_points=N, device=local_rank, **args_map)
    for k in range(min(6, F)):  # warm-up: the first frames are gated (PCR:199), the ICP kernels first run on frame 3;
        lm.process_new_scan(scans[k])  # their code objects load lazily.  The sequence restarts below
    lm.close()
    lm = Laser_mapping(scan_points=N, device=local_rank, **args_map)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    accepted, errs = 0, []
    for k in range(F):
        # (the feature node runs beside the mapping node in the reference: frame k + 1 is extracted while frame k registers)
        accepted += lm.process_new_scan(scans[k], next_xyzi=(scans[k + 1] if k + 1 < F and not args.no_prefetch else None))
        errs.append(synth.pose_error(lm.pose, truth[k]))
    lm.sync()  # (the cell maps' service thread has appended every frame: inside the timed region)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t1 = time.perf_counter()
    cell_maps = lm.keep_cell_maps
    if cell_maps:
        views = [lm.history.cell_map(k).device_view(local_rank) for k in (0, 1)]  # device-resident: (points (n, 4), cell keys (n,)) per kind
        cells_per_kind = [int(lm.history.cell_map(k).stats()[0]) for k in (0, 1)]
        counts_kind = [[int(v[0].shape[0])] for v in views]
        bytes_per_point = 24
    else:
        sub = torch.cat([lm.history.map_cloud_device(0), lm.history.map_cloud_device(1)], 0)
        counts_kind, cells_per_kind, bytes_per_point = [[int(sub.shape[0])]], None, 16
    merged_cells = None
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        world = dist.get_world_size()  # what RCCL saw
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if cell_maps:
            merged = [gather_cell_maps(v[0], v[1], dist) for v in views]
            counts_kind = [m[3] for m in merged]
            merged_cells = [int(m[2].shape[0]) - 1 for m in merged]
        else:
            counts_kind = [gather_submaps(sub, dist)[1]]
        torch.cuda.synchronize()
    t_gather = time.perf_counter() - t1
    counts = [int(sum(c[r] for c in counts_kind)) for r in range(len(counts_kind[0]))]
    result = {
        "metric": "frames_per_s", "value": round(F * world / elapsed, 2), "unit": "frames/s (sequential mapping loop with local map growth, all ranks)",
        "n_gpus": world, "frames_per_sequence": F, "ms_per_frame": round(1e3 * elapsed / F, 3), "scaling": "weak",
        "config": {"workload": "C4 unit: one sequence per GPU, 24k-pt scans, " + ("cell-map" if args.matching_mode else "history") + " match buffer (local growth), VoxelGrid "
                               f"{args.line_res}/{args.plane_res}, 10 ICP iters max", "history": args.history},
        "accepted": accepted, "final_drift_m": float(errs[-1][0]), "final_drift_rad": float(errs[-1][1]),
        "max_drift_m": float(max(e[0] for e in errs)), "submap_points_per_rank": counts,
        "submap": ({"what": "corner + surface cell maps of the whole sequence (cell key + point, 24 B per point)", "cells_this_rank": {"corner": cells_per_kind[0], "surface": cells_per_kind[1]},
                    "points_this_rank": {"corner": counts_kind[0][rank if rank < len(counts_kind[0]) else 0], "surface": counts_kind[1][rank if rank < len(counts_kind[1]) else 0]},
                    "cells_of_the_union": ({"corner": merged_cells[0], "surface": merged_cells[1]} if merged_cells else None)}
                   if cell_maps else {"what": "history match buffer (16 B per point)"}),
        "gather_s": round(t_gather, 4),
        "gather_gb_per_s_received_per_rank": round((sum(counts) - counts[rank if rank < len(counts) else 0]) * bytes_per_point / max(t_gather, 1e-9) / 1e9, 3) if len(counts) > 1 else None,
        "match_buffer": {"corner": lm.map_sizes[0], "surface": lm.map_sizes[1]},
        "ms_per_frame_by_stage": dict(zip(("extract_register", "history_add", "match_buffer_refresh"), [round(1e3 * float(v) / F, 3) for v in lm.stage_s[:3]])),
        "icp_iterations_last_frame": int(lm.last_report.icp_iterations),
        "next_frame_extracted_during_registration": not args.no_prefetch,
    }
    cyc = [int(v) for v in lm.reg.debug_cycles(0)]
    if any(cyc):  # only the -DLL_SOLVE_TIMING build (LOAM_LIVOX_LIB=...timing.so) fills these
        result["solver_phase_cycles_last_frame"] = cyc
        result["lm_iterations_last_frame"] = int(lm.last_report.lm_iterations_total)
        result["blocks_last_frame"] = int(lm.last_report.n_blocks_last)
    if rank == 0 and args.cpu_frames > 0:
        from oracle.orc_mapping import LaserMapping  # the checker, timed beside the device loop
        args_cpu = {k_: v_ for k_, v_ in args_map.items() if k_ != "keep_cell_maps"}  # (the sequential loop never reads the cell maps)
        om = LaserMapping(**args_cpu)
        lm2 = Laser_mapping(scan_points=N, device=local_rank, **args_cpu)
        tb = time.perf_counter()
        worst = (0.0, 0.0)
        n_cpu = min(args.cpu_frames, F)
        t_cpu = 0.0
        for k in range(n_cpu):
            tb = time.perf_counter()
            om.process_new_scan(scans[k])
            t_cpu += time.perf_counter() - tb
            lm2.process_new_scan(scans[k])
            e = synth.pose_error(lm2.pose, om.pose)
            worst = (max(worst[0], e[0]), max(worst[1], e[1]))
        result["cpu_baseline"] = {"value": round(n_cpu / t_cpu, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                                  "sample": f"first {n_cpu} frames of rank 0's sequence (k-d tree rebuilds included: they are part of the refresh)"}
        result["parity_vs_cpu"] = {"max_pose_err_m": worst[0], "max_pose_err_rad": worst[1]}
        lm2.close()
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
