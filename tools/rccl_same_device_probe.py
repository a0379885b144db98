#!/usr/bin/env python3
"""Can two RCCL ranks share ONE GPU on this image?  (If so, the N > 1 nccl path of dist.gather_rows / count_ranks can run on a one-GPU
box; if RCCL refuses — "Duplicate GPU detected" — the gloo rehearsal stays the only one.)  Starts its own two ranks (dist.launch_ranks),
both on device 0, backend nccl, bounded by a timeout.  usage (GPU box, repo root): timeout -k 10 150 python tools/rccl_same_device_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smplpp_amd import dist as D  # noqa: E402

if D.launch_plan(2 if "RANK" not in os.environ else int(os.environ["WORLD_SIZE"])) == "spawn":
    raise SystemExit(D.launch_ranks([sys.executable, os.path.abspath(__file__)], 2, timeout=120))

import torch  # noqa: E402

rank, world, _ = D.env_rank_world()
torch.cuda.set_device(0)
os.environ["LOCAL_RANK"] = "0"
D.init_process_group("nccl")
c = D.count_ranks()
lo, hi = D.shard_range(11, rank, world)
local = (torch.arange(lo, hi, dtype=torch.float32, device="cuda")[:, None] * torch.tensor([[1.0, 10.0]], device="cuda"))
full = D.gather_rows(local, 11)
sp = D.timed_region(lambda: None, torch.cuda.synchronize)
if rank == 0:
    want = torch.arange(11, dtype=torch.float32, device="cuda")[:, None] * torch.tensor([[1.0, 10.0]], device="cuda")
    print("RCCL two ranks on one device:", c, "gather ok:", bool(torch.equal(full, want)), "region spread:", sp)
