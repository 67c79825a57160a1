"""Multi-GPU driver for batched offline map building (BASELINE config C4).

Per-scan odometry does not shard (each scan needs the previous pose and the resident map): "replicas only".
What shards trivially is a set of independent sub-sequences: one process per GPU, each rank registers its own
scans against its own resident map with NO data-path collective; the only exchange step is the gather of the
per-rank sub-maps at the end.  The sub-maps are variable-length and stay where they were produced: each rank's
contribution is a DEVICE tensor (assembled by ll_cloud_transform_fe_device / ll_history_map_cloud_device, no host
copy), the counts go round in one small all_gather, and the points travel as one grouped batch of point-to-point
sends -- every pair of GPUs of an MI355X node has its own xGMI link, so the pairwise pattern uses all 7 links of a
GPU at once, moves exactly the bytes that exist (no padding to the largest rank), and lands each block at its final
offset in the receiver's output.  RCCL when the backend is "nccl", gloo in the CPU tests.

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
    """All-gather of variable-length point arrays.  local_pts: torch tensor (n, C) on the rank's device (GPU for nccl,
    CPU for gloo); it is sent from where it lies.  Returns (concatenation in rank order on the same device, counts)."""
    import torch
    import torch.distributed as td
    dist = dist or td
    world, rank = dist.get_world_size(), dist.get_rank()
    local_pts = local_pts.contiguous()
    n = torch.tensor([local_pts.shape[0]], dtype=torch.int64, device=local_pts.device)
    counts_t = torch.zeros(world, dtype=torch.int64, device=local_pts.device)
    dist.all_gather_into_tensor(counts_t, n)
    counts = [int(c) for c in counts_t.tolist()]
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    out = torch.empty((int(offs[-1]), local_pts.shape[1]), dtype=local_pts.dtype, device=local_pts.device)
    out[offs[rank]:offs[rank + 1]] = local_pts
    ops = []
    for step in range(1, world):  # pairwise exchange: in step s rank r sends to r+s and receives from r-s
        dst, src = (rank + step) % world, (rank - step) % world
        if counts[rank] > 0:
            ops.append(dist.P2POp(dist.isend, local_pts, dst))
        if counts[src] > 0:
            ops.append(dist.P2POp(dist.irecv, out[offs[src]:offs[src + 1]], src))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return out, counts


def cell_keys(cell_ijk) -> np.ndarray:
    """the device's 64-bit cell key of (i, j, k) cell indices (ll_cellmap_core.h cell_pack: 21 bits per axis, i most significant, so key
    order is the lexicographic order of the indices), as int64"""
    c = np.asarray(cell_ijk, np.int64).reshape(-1, 3) + (1 << 20)
    return (c[:, 0] << 42) | (c[:, 1] << 21) | c[:, 2]


def gather_cell_maps(points, keys, dist=None):
    """The exchange step of batched offline map building as BASELINE config C4 words it: a gather of the ranks' CELL MAPS
    (m_pt_cell_map_corners / m_pt_cell_map_planes, laser_mapping.hpp:274-275, fed by every registered frame at :1492-1493;
    cell_map_keyframe.hpp:477-672).  A rank's map is what the device holds: `points` (n, 4) float32 in (cell, insertion) order and
    `keys` (n,) int64, the cell key of every point (api.Cell_map.device_view).  One exchange: points and keys travel together as
    24-byte rows through gather_submaps (counts all-gather + grouped point-to-point sends), then the union is put back into cell-map
    layout -- a stable sort by cell key, so a cell that several ranks saw holds rank 0's points first, each rank's in insertion order.
    Returns (merged points (N, 4), merged keys (N,), first point of each distinct cell (C + 1,), per-rank point counts)."""
    import torch
    n = int(points.shape[0])
    assert points.shape[1] == 4 and keys.shape[0] == n
    rows = torch.cat([points.contiguous().view(torch.float32), keys.contiguous().view(torch.float32).reshape(n, 2)], 1) if n else \
        torch.zeros((0, 6), dtype=torch.float32, device=points.device)
    allrows, counts = gather_submaps(rows, dist)
    pts = allrows[:, :4].contiguous()
    k64 = allrows[:, 4:6].contiguous().view(torch.int64).reshape(-1)
    order = torch.sort(k64, stable=True).indices
    pts, k64 = pts[order], k64[order]
    if len(k64):
        head = torch.ones(len(k64), dtype=torch.bool, device=k64.device)
        head[1:] = k64[1:] != k64[:-1]
        cell_start = torch.cat([torch.nonzero(head).reshape(-1), torch.tensor([len(k64)], device=k64.device)])
    else:
        cell_start = torch.zeros(1, dtype=torch.int64, device=k64.device)
    return pts, k64, cell_start, counts


class DeviceHandles:
    """The three device handles a rank's share runs on (one GPU): resident map, batched extractor, batched registrar."""

    def __init__(self, corner_map: np.ndarray, surf_map: np.ndarray, device: int, scan_points: int, batch: int, icp_iters: int = 10):
        import torch
        from .api import Livox_laser, Map_buffer, Point_cloud_registration
        self.torch_device = torch.device(f"cuda:{device}")
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

    def register_batch(self, scans: np.ndarray, poses_init: np.ndarray):
        """extract + register one batch; the selections stay resident in the extractor for append_submap"""
        n = scans.shape[0]
        self.fe.upload(scans, np.full(n, 1.0))
        self.fe.extract_batch(n)
        self.fe.resolve()
        self.fe.select_batch(n, 0)
        res, pc, _, _ = self.reg.solve_batch_fe(self.map, self.fe, n, poses_init, poses_init)
        return res, pc

    def append_submap(self, n: int, accept: np.ndarray, poses: np.ndarray, out, n_used: int) -> int:
        """surface features of the accepted scans of the batch just registered -> map frame -> rows of `out` (device)"""
        from .api import Map_buffer
        return self.reg.append_to_submap_device(self.fe, n, Map_buffer.SURF, accept, poses, out, n_used)


class SequenceRunner:
    """One rank's share of the batched job: registers its scans (independent units) against the rank's resident map
    and returns the poses plus the sub-map (accepted scans' surface features moved to the map frame), which is built
    on the device from the batch's resident selections: one device-to-device transform per accepted scan, no second
    upload or extraction and no device-to-host copy of the points.

    `handles` is anything with torch_device, register_batch(scans, poses_init) -> (results, poses) and
    append_submap(n, accept, poses, out, n_used) -> n_used'; DeviceHandles on a GPU, a stub in the gloo test."""

    def __init__(self, handles, scan_points: int, batch: int):
        self.h, self.scan_points, self.batch = handles, scan_points, batch

    @classmethod
    def on_device(cls, corner_map, surf_map, device: int, scan_points: int, batch: int, icp_iters: int = 10):
        return cls(DeviceHandles(corner_map, surf_map, device, scan_points, batch, icp_iters), scan_points, batch)

    def run(self, scans: np.ndarray, poses_init: np.ndarray):
        """scans (S, N, 4) float32, poses_init (S, 7).  Returns (results (S,), poses (S, 7), submap torch tensor (n, 4)
        on handles.torch_device)."""
        import torch
        S = scans.shape[0]
        poses = np.zeros((S, 7))
        results = np.zeros(S, np.int32)
        sub = torch.empty((max(1, S * self.scan_points), 4), dtype=torch.float32, device=self.h.torch_device)
        used = 0
        for lo in range(0, S, self.batch):
            hi = min(S, lo + self.batch)
            res, pc = self.h.register_batch(scans[lo:hi], poses_init[lo:hi])
            poses[lo:hi], results[lo:hi] = pc, res
            used = self.h.append_submap(hi - lo, res, pc, sub, used)
        return results, poses, sub[:used]


def run_sharded(runner: SequenceRunner, scans: np.ndarray, poses_init: np.ndarray, dist=None):
    """The whole multi-GPU job from one rank's point of view: take this rank's contiguous share of the S independent
    scans, run it, then exchange -- sub-maps by gather_submaps, the (S, 8) result/pose table by the same gather.
    Returns (results (S,), poses (S, 7), merged sub-map tensor, per-rank point counts), identical on every rank."""
    import torch
    import torch.distributed as td
    dist = dist or td
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = shard_range(scans.shape[0], rank, world)
    res, poses, sub = runner.run(scans[mine.start:mine.stop], poses_init[mine.start:mine.stop])
    table = torch.from_numpy(np.concatenate([res[:, None].astype(np.float64), poses], axis=1)).to(sub.device)
    table_all, _ = gather_submaps(table, dist)
    merged, counts = gather_submaps(sub, dist)
    table_all = table_all.cpu().numpy()
    return table_all[:, 0].astype(np.int32), table_all[:, 1:], merged, counts
