"""-m gpu: the device-resident pieces of the multi-GPU path on the one GPU a test box has -- the sub-map
assembled from an extractor's resident selections (ll_cloud_transform_fe_device), the match-buffer clouds read where
they lie (ll_history_map_cloud_device), and gather_submaps over RCCL with a world of one.  The sharding logic itself
is covered by the world-size-2 gloo tests (tests/test_dist_gloo.py)."""
import os
import socket

import numpy as np
import pytest

from loam_livox_amd import synth
from loam_livox_amd.api import History_buffer, Livox_laser, Map_buffer, Point_cloud_registration
from loam_livox_amd.capi import LoamLivoxError
from loam_livox_amd.multigpu import SequenceRunner, gather_submaps, run_sharded

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small(gpu_lib):
    world, corner, surf = synth.make_maps(200_000)
    scans = [synth.make_scan(world, 500 + k) for k in range(5)]
    return dict(world=world, corner=corner, surf=surf, scans=scans)


def test_submap_from_resident_selections_equals_per_scan_host_path(small):
    import torch
    N = 24000
    xyzi = np.stack([s.xyzi for s in small["scans"]])
    inits = np.stack([s.pose_init for s in small["scans"]])
    runner = SequenceRunner.on_device(small["corner"], small["surf"], 0, N, batch=3)
    res, poses, sub = runner.run(xyzi, inits)
    assert sub.is_cuda and list(res) == [1, 1, 1, 1, 1]
    # the route round 1 took: re-extract every accepted scan on its own, copy its features out, transform on request
    reg = Point_cloud_registration(max_scans=1, max_features=N)
    want = []
    for b in range(len(xyzi)):
        fe = Livox_laser(max_points=N, piecewise_number=1)
        fe.upload(xyzi[b:b + 1], np.full(1, 1.0))  # slot 0 of a one-scan batch: the same time base as the runner's slots
        fe.extract_batch(1)
        fe.resolve()
        sp = fe.splits()
        g = fe.get_features(float(sp["piece_start"][0]), float(sp["piece_end"][0]))  # the window select_batch(n, 0) uses
        want.append(reg.pointcloudAssociateToMap(g["pc_surface"], poses[b]))
        fe.close()
    reg.close()
    assert np.array_equal(sub.cpu().numpy(), np.concatenate(want))
    # a rejected scan contributes nothing: the last batch (scans 3 and 4) is still resident in the extractor
    part = torch.zeros((N, 4), dtype=torch.float32, device="cuda:0")
    used = runner.h.append_submap(2, np.array([0, 1], np.int32), poses[3:5], part, 7)
    assert used == 7 + len(want[4]) and np.array_equal(part[7:used].cpu().numpy(), want[4]) and float(part[:7].abs().sum()) == 0.0
    # capacity is checked before anything is written
    tiny = torch.zeros((10, 4), dtype=torch.float32, device="cuda:0")
    with pytest.raises(LoamLivoxError):
        runner.h.append_submap(1, np.array([1], np.int32), poses[-1:], tiny, 0)
    assert float(tiny.abs().sum()) == 0.0


def test_history_clouds_on_device_equal_host_copies(small):
    N = 24000
    fe = Livox_laser(max_points=N, piecewise_number=1)
    h = History_buffer(4, N, 0.1, 0.15)
    m = Map_buffer()
    for k, sc in enumerate(small["scans"][:3]):
        fe.extract_laser_features(sc.xyzi, 1.0)
        fe.get_features(0.0, 1.0)  # the selection add_fe reads
        h.add_fe(fe, 0, sc.pose_true, 0.0, 0.0)
    h.refresh(m)
    for kind in (0, 1):
        dev = h.map_cloud_device(kind)
        assert dev.is_cuda and np.array_equal(dev.cpu().numpy(), h.map_cloud(kind)) and len(dev) > 0
    fe.close(); h.close(); m.close()


def test_gather_over_rccl_world_of_one(small):
    import torch
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        pts = torch.arange(40, dtype=torch.float32, device="cuda:0").reshape(10, 4)
        out, counts = gather_submaps(pts)
        assert counts == [10] and torch.equal(out, pts) and out.is_cuda
        xyzi = np.stack([s_.xyzi for s_ in small["scans"][:2]])
        inits = np.stack([s_.pose_init for s_ in small["scans"][:2]])
        runner = SequenceRunner.on_device(small["corner"], small["surf"], 0, 24000, batch=2)
        res, poses, merged, counts = run_sharded(runner, xyzi, inits)
        r1, p1, m1 = runner.run(xyzi, inits)
        assert np.array_equal(res, r1) and np.array_equal(poses, p1) and torch.equal(merged, m1) and counts == [len(m1)]
        # the cell-map gather (BASELINE config C4's exchange step) on a device cell map: a world of one returns the map unchanged
        from loam_livox_amd.api import Cell_map
        from loam_livox_amd.multigpu import gather_cell_maps
        rng = np.random.default_rng(3)
        cm = Cell_map(1 << 14, 1.0)
        cm.append_cloud(np.c_[rng.uniform(-5, 5, (2500, 3)), np.zeros(2500)].astype(np.float32))
        p_dev, k_dev = cm.device_view(0)
        mp, mk, cell_start, counts = gather_cell_maps(p_dev, k_dev)
        assert counts == [2500] and torch.equal(mp, p_dev) and torch.equal(mk, k_dev) and mp.is_cuda
        assert np.array_equal(cell_start.cpu().numpy(), cm.dump()[2].astype(np.int64))
        cm.close()
    finally:
        dist.destroy_process_group()


def test_cell_map_device_view_equals_the_dump(gpu_lib):
    """ll_cellmap_device_view (the input of multigpu.gather_cell_maps on a GPU): the points and per-point cell keys read where they lie equal
    ll_cellmap_dump's host copies (gather_cell_maps over an RCCL world of one: test_gather_over_rccl_world_of_one)"""
    import torch
    from loam_livox_amd.api import Cell_map
    from loam_livox_amd.multigpu import cell_keys
    rng = np.random.default_rng(21)
    cm = Cell_map(1 << 14, 1.0)
    for k in range(3):
        cm.append_cloud(np.c_[rng.uniform(-6, 6, (1500 + 300 * k, 3)), np.zeros(1500 + 300 * k)].astype(np.float32))
    xyz, ijk, start, _ = cm.dump()
    pts, keys = cm.device_view(0)
    assert pts.is_cuda and keys.is_cuda and pts.shape == (len(xyz), 4) and keys.dtype == torch.int64
    assert np.array_equal(pts.cpu().numpy()[:, :3], xyz)
    assert np.array_equal(keys.cpu().numpy(), np.repeat(cell_keys(ijk), np.diff(start)))
    cm.close()


def test_knn5_on_device_resident_queries_equals_the_host_call(gpu_lib, small_world):
    """ll_map_knn5_device: the same lists as ll_map_knn5, nothing crossing PCIe, a positive kernel time"""
    import torch
    from loam_livox_amd.api import Map_buffer
    m = Map_buffer()
    m.setInputCloud(Map_buffer.SURF, small_world["surf"])
    rng = np.random.default_rng(6)
    q = (small_world["surf"][rng.choice(len(small_world["surf"]), 20000, replace=False), :3] + rng.normal(0, 0.05, (20000, 3))).astype(np.float32)
    hi, hd = m.nearestKSearch(Map_buffer.SURF, q, 50.0)
    dq = torch.from_numpy(q).cuda()
    di = torch.empty((len(q), 5), dtype=torch.int32, device="cuda")
    dd = torch.empty((len(q), 5), dtype=torch.float32, device="cuda")
    ms = m.nearestKSearch_device(Map_buffer.SURF, dq, 50.0, di, dd)
    assert ms > 0.0 and np.array_equal(di.cpu().numpy(), hi) and np.array_equal(dd.cpu().numpy(), hd)
    assert m.cells(Map_buffer.SURF) > 1000
    m.close()
