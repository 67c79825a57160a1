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
