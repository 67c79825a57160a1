"""Multi-GPU driver for batched offline map building (BASELINE config C4).

Per-scan odometry does not shard (each scan needs the previous pose and the resident map): "replicas only".
What shards trivially is a set of independent sub-sequences: one process per GPU, each rank registers its own
scans against its own resident map with NO data-path collective; the only exchange step is the gather of the
per-rank sub-maps at the end (variable-length, so: all_gather of counts, then a padded all_gather of the point
arrays -- RCCL over xGMI when the backend is "nccl", gloo in the CPU tests).

The reference has no analogue of the collective (closest: Mapping_refine::refine_mapping concatenating keyframe
clouds, source/ceres_pose_graph_3d.hpp:503-538).
"""
from __future__ import annotations

import numpy as np


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced partition of n_items units over `world` ranks (first ranks take the remainder)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def gather_submaps(local_pts, dist=None):
    """All-gather variable-length point arrays.  local_pts: torch tensor (n, C) on the rank's device (GPU for nccl,
    CPU for gloo).  Returns (concatenated tensor in rank order, counts list)."""
    import torch
    import torch.distributed as td
    dist = dist or td
    world = dist.get_world_size()
    n = torch.tensor([local_pts.shape[0]], dtype=torch.int64, device=local_pts.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    padded = torch.zeros((cap, local_pts.shape[1]), dtype=local_pts.dtype, device=local_pts.device)
    padded[: local_pts.shape[0]] = local_pts
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0), counts


class SequenceRunner:
    """One rank's share of the batched job: registers its scans (independent units) against the rank's resident map
    and returns the poses plus the sub-map (accepted scans' features moved to the map frame)."""

    def __init__(self, corner_map: np.ndarray, surf_map: np.ndarray, device: int, scan_points: int, batch: int, icp_iters: int = 10):
        from .api import Livox_laser, Map_buffer, Point_cloud_registration
        self.map = Map_buffer(device=device)
        self.map.setInputCloud(Map_buffer.CORNER, corner_map)
        self.map.setInputCloud(Map_buffer.SURF, surf_map)
        self.fe = Livox_laser(max_points=scan_points, max_scans=batch, device=device, piecewise_number=1)
        self.reg = Point_cloud_registration(max_scans=batch, max_features=scan_points, device=device)
        p = self.reg.params
        p.icp_max_iterations, p.ceres_max_iterations, p.force_all_iterations = icp_iters, 20, 0
        p.para_max_angular_rate, p.para_max_speed, p.max_final_cost = 20.0, 0.3, 1000.0
        p.current_frame_index, p.mapping_init_accumulate_frames = 100, 50
        p.maximum_allow_residual_block = scan_points
        self.batch = batch

    def run(self, scans: np.ndarray, poses_init: np.ndarray):
        """scans (S, N, 4) float32, poses_init (S, 7).  Returns (results, poses, submap_xyzi)."""
        S = scans.shape[0]
        poses = np.zeros((S, 7))
        results = np.zeros(S, np.int32)
        sub = []
        for lo in range(0, S, self.batch):
            hi = min(S, lo + self.batch)
            n = hi - lo
            self.fe.upload(scans[lo:hi], np.full(n, 1.0))
            self.fe.extract_batch(n)
            self.fe.resolve()
            self.fe.select_batch(n, 0)
            res, pc, _, _ = self.reg.solve_batch_fe(self.map, self.fe, n, poses_init[lo:hi], poses_init[lo:hi])
            poses[lo:hi], results[lo:hi] = pc, res
            for b in range(n):
                if res[b]:
                    self.fe.upload(scans[lo + b:lo + b + 1], np.full(1, 1.0))  # slot 0 view for the per-scan accessors
                    self.fe.extract_batch(1)
                    g = self.fe.get_features(0.0, 1.0)
                    sub.append(self.reg.pointcloudAssociateToMap(g["pc_surface"], pc[b]))
        submap = np.concatenate(sub, axis=0) if sub else np.zeros((0, 4), np.float32)
        return results, poses, submap
