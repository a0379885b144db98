"""The N>1 path on CPU: world_size-2 gloo processes shard independent units and gather them back in order."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from smplpp_amd import dist as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("total,world", [(1024, 8), (64, 8), (10, 4), (3, 4), (0, 2), (3163, 8)])
def test_shards_tile_the_range(total, world):
    edges = [D.shard_range(total, r, world) for r in range(world)]
    assert edges[0][0] == 0 and edges[-1][1] == total
    for a, b in zip(edges, edges[1:]):
        assert a[1] == b[0]
    sizes = D.shard_sizes(total, world)
    assert sum(sizes) == total and max(sizes) - min(sizes) <= 1


WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch
from smplpp_amd import dist as D
d = D.init_process_group("gloo")
rank, world, _ = D.env_rank_world()
total = 11
lo, hi = D.shard_range(total, rank, world)
local = torch.arange(lo, hi, dtype=torch.float32)[:, None] * torch.tensor([[1.0, 10.0]])
D.barrier()
full = D.gather_rows(local, total)
t = D.max_over_ranks(float(rank + 1))
if rank == 0:
    want = torch.arange(total, dtype=torch.float32)[:, None] * torch.tensor([[1.0, 10.0]])
    assert full is not None and torch.equal(full, want), full
    assert t == float(world)
    print("OK")
else:
    assert full is None
d.destroy_process_group()
"""


def test_two_rank_gloo_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "OK" in outs[0]
