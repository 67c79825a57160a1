"""World-size-2 gloo test (CPU) of the only multi-process logic on the path: the contiguous sharding of independent
scan sub-sequences and the variable-length sub-map gather that ends a batched run (BASELINE config C4)."""
import os
import socket
import subprocess
import sys

import numpy as np

from loam_livox_amd.multigpu import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from loam_livox_amd.multigpu import gather_submaps, shard_range
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
units = list(shard_range(11, rank, world))
rng = np.random.default_rng(100 + rank)
n = 5 + 7 * rank                      # ragged sizes; rank 0 also covers the "smaller than the pad" case
local = torch.from_numpy(rng.normal(size=(n, 4)).astype(np.float32))
local[:, 3] = rank
allpts, counts = gather_submaps(local)
assert counts == [5 + 7 * r for r in range(world)], counts
off = sum(counts[:rank])
assert torch.equal(allpts[off:off + n], local)
for r in range(world):
    o = sum(counts[:r])
    assert torch.all(allpts[o:o + counts[r], 3] == r)
# empty contribution from one rank
empty = torch.zeros((0, 4)) if rank == 1 else local
allpts2, counts2 = gather_submaps(empty)
assert counts2[1] == 0 and allpts2.shape[0] == counts2[0]
# barrier + max-over-ranks timing reduction used by bench.py
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert t.item() == world
print("rank", rank, "units", units, "ok")
dist.destroy_process_group()
"""


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 2000, 16001):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in shard_range(n, r, world)]
            assert got == list(range(n))
            sizes = [len(shard_range(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_gather_submaps_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(free_port()), str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


RUNNER_WORKER = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from loam_livox_amd.multigpu import SequenceRunner, run_sharded, shard_range


class StubHandles:
    '''stands in for DeviceHandles (map + extractor + registrar of one GPU) on the CPU: "registration" returns the
    initial pose shifted by the scan's mean point and rejects scans whose first intensity is negative; the "features"
    of a scan are its first 3 + (index %% 4) points.  Deterministic per scan, so any sharding must reproduce it.'''
    torch_device = torch.device("cpu")

    def register_batch(self, scans, poses_init):
        self.last = scans
        res = (scans[:, 0, 3] >= 0).astype(np.int32)
        poses = poses_init.copy()
        poses[:, 4:] += scans[:, :, :3].mean(1)
        return res, poses

    def append_submap(self, n, accept, poses, out, n_used):
        for b in range(n):
            if accept[b]:
                k = 3 + int(self.last[b, 0, 3]) %% 4
                pts = torch.from_numpy(self.last[b, :k].copy())
                pts[:, :3] += torch.from_numpy(poses[b, 4:].astype(np.float32))
                out[n_used:n_used + k] = pts
                n_used += k
        return n_used


def job(S):
    rng = np.random.default_rng(77)
    scans = rng.normal(size=(S, 16, 4)).astype(np.float32)
    scans[:, 0, 3] = np.arange(S)
    scans[S // 3, 0, 3] = -1.0            # one rejected scan
    poses = np.tile(np.array([0, 0, 0, 1, 0, 0, 0], np.float64), (S, 1))
    poses[:, 4] = np.arange(S)
    return scans, poses


dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
for S, batch in ((11, 4), (2, 4), (1, 4)):   # ragged shares, a share smaller than a batch, and a rank with nothing to do
    scans, poses0 = job(S)
    res, poses, merged, counts = run_sharded(SequenceRunner(StubHandles(), 16, batch), scans, poses0)
    # the single-process answer
    r1, p1, m1 = SequenceRunner(StubHandles(), 16, batch).run(scans, poses0)
    assert np.array_equal(res, r1) and np.array_equal(poses, p1), (S, rank)
    assert torch.equal(merged, m1), (S, rank)
    mine = shard_range(S, rank, world)
    assert counts[rank] == sum(3 + i %% 4 for i in mine if i != S // 3), (counts, S, rank)
print("rank", rank, "runner ok")
dist.destroy_process_group()
"""


def test_sequence_runner_sharding_world2_gloo(tmp_path):
    """SequenceRunner + run_sharded over two gloo ranks with a stubbed registrar: shares, batching inside a share,
    rejected scans, the device-side sub-map append and both gathers reproduce the single-process run exactly."""
    script = tmp_path / "runner_worker.py"
    script.write_text(RUNNER_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(free_port()), str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("runner ok") == 2


CELLMAP_WORKER = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from loam_livox_amd.multigpu import cell_keys, gather_cell_maps
from oracle.orc_cellmap import CellMap   # the checker: real cell maps of different sizes, one per rank

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()


def rank_map(r):
    # what rank r's sequence left in its cell map: five clouds over an area that overlaps the other rank's
    rng = np.random.default_rng(40 + r)
    m = CellMap(1.0)
    for k in range(5):
        n = 300 + 450 * r + 40 * k
        m.append(np.c_[rng.uniform(-4 + 3 * r, 5 + 3 * r, (n, 3)), np.zeros(n)].astype(np.float32))
    return m


def as_rows(m):
    xyz, ijk, start, _ = m.dump()
    pts = np.c_[xyz, np.zeros(len(xyz), np.float32)].astype(np.float32)
    keys = np.repeat(cell_keys(ijk), np.diff(start))
    return pts, keys


pts, keys = as_rows(rank_map(rank))
mp, mk, cell_start, counts = gather_cell_maps(torch.from_numpy(pts), torch.from_numpy(keys))
sizes = [len(as_rows(rank_map(r))[1]) for r in range(world)]
assert counts == sizes and sizes[0] != sizes[1], (counts, sizes)
# the union in cell-map layout = the cell map that received rank 0's stored points, then rank 1's (a cell's points in insertion order)
union = CellMap(1.0)
for r in range(world):
    union.append(as_rows(rank_map(r))[0])
uxyz, uijk, ustart, _ = union.dump()
assert np.array_equal(mp.numpy()[:, :3], uxyz) and np.array_equal(mk.numpy(), np.repeat(cell_keys(uijk), np.diff(ustart)))
assert np.array_equal(cell_start.numpy(), ustart.astype(np.int64))
shared = np.intersect1d(as_rows(rank_map(0))[1], as_rows(rank_map(1))[1])
assert len(shared) > 10   # the two maps really overlap: cells with points of both ranks
# a rank whose map is empty
e_pts, e_keys = (torch.zeros((0, 4)), torch.zeros(0, dtype=torch.int64)) if rank == 0 else (torch.from_numpy(pts), torch.from_numpy(keys))
mp2, mk2, cs2, counts2 = gather_cell_maps(e_pts, e_keys)
assert counts2[0] == 0 and len(mk2) == counts2[1] and bool(torch.all(mk2[1:] >= mk2[:-1]))
print("rank", rank, "cell maps ok", counts)
dist.destroy_process_group()
"""


def test_gather_cell_maps_world2_gloo(tmp_path):
    """the exchange step of C4 as BASELINE words it (a gather of cell maps): two ranks with real cell maps of different sizes over
    overlapping areas; the union comes back in cell-map layout, equal to the map that appended both ranks' points"""
    script = tmp_path / "cellmap_worker.py"
    script.write_text(CELLMAP_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(free_port()), str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("cell maps ok") == 2


CELLMAP8_WORKER = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from loam_livox_amd.multigpu import cell_keys, gather_cell_maps, gather_submaps
from oracle.orc_cellmap import CellMap

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
assert world == 8
SIZES = [700, 0, 1900, 130, 0, 2600, 1, 940]   # ragged, two ranks with nothing, one with a single point


def rank_rows(r):
    rng = np.random.default_rng(900 + r)
    m = CellMap(1.0)
    n = SIZES[r]
    if n:
        m.append(np.c_[rng.uniform(-6 + 2 * r, 4 + 2 * r, (n, 3)), np.zeros(n)].astype(np.float32))
    xyz, ijk, start, _ = m.dump()
    return np.c_[xyz, np.zeros(len(xyz), np.float32)].astype(np.float32).reshape(-1, 4), np.repeat(cell_keys(ijk), np.diff(start)).astype(np.int64)


pts, keys = rank_rows(rank)
mp, mk, cell_start, counts = gather_cell_maps(torch.from_numpy(pts), torch.from_numpy(keys))
assert counts == SIZES, counts
union = CellMap(1.0)
for r in range(world):
    p, _ = rank_rows(r)
    if len(p):
        union.append(p)
uxyz, uijk, ustart, _ = union.dump()
assert np.array_equal(mp.numpy()[:, :3], uxyz) and np.array_equal(mk.numpy(), np.repeat(cell_keys(uijk), np.diff(ustart)))
assert np.array_equal(cell_start.numpy(), ustart.astype(np.int64))
# every rank holds the same union (checksum of checksums over the ranks)
chk = torch.tensor([float(mp.double().sum()), float(mk.double().sum())], dtype=torch.float64)
allchk = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
dist.all_gather(allchk, chk)
assert all(torch.equal(c, allchk[0]) for c in allchk)
# the plain sub-map gather with the same ragged sizes, every rank empty but one, and everyone empty
out, c2 = gather_submaps(torch.from_numpy(pts))
assert c2 == SIZES and out.shape[0] == sum(SIZES)
lone = torch.from_numpy(pts) if rank == 5 else torch.zeros((0, 4))
out3, c3 = gather_submaps(lone)
assert c3 == [0, 0, 0, 0, 0, SIZES[5], 0, 0] and torch.equal(out3, torch.from_numpy(rank_rows(5)[0]))
out4, c4 = gather_submaps(torch.zeros((0, 4)))
assert c4 == [0] * 8 and out4.shape == (0, 4)
print("rank", rank, "cell maps of eight ok")
dist.destroy_process_group()
"""


def test_gather_cell_maps_world8_ragged_and_empty_ranks_gloo(tmp_path):
    """eight ranks (the node the driver measures on) with ragged cell maps, two of them empty, one holding a single point: counts,
    the union in cell-map layout against the oracle's cell map fed rank by rank, the same union on every rank; and the sub-map
    gather when only one rank / no rank has anything to send"""
    script = tmp_path / "cellmap8_worker.py"
    script.write_text(CELLMAP8_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
                          "--master-port", str(free_port()), str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("cell maps of eight ok") == 8
