"""Multi-GPU: frames (FK) or whole sequences/restarts (IK) are independent, so a job shards embarrassingly — one
process per GPU, contiguous blocks per rank, NO collective inside the compute — and the only exchange is the final
gather of results (RCCL over xGMI: `torch.distributed` backend "nccl" is RCCL on ROCm; "gloo" on CPU for tests).

The reference has no counterpart (single process, device index 0: node/node.cpp:372); SURVEY.md §8(e) is the spec.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import numpy as np


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment; (0, 1, 0) when not launched distributed."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of `total` independent units for `rank`; sizes differ by at most one and the
    blocks tile [0, total) in rank order (strong scaling: total fixed)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(total: int, world: int) -> List[int]:
    return [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]


def init_process_group(backend: str | None = None):
    """Initialise torch.distributed from the torchrun environment (MASTER_ADDR defaults to 127.0.0.1)."""
    import torch
    import torch.distributed as dist

    rank, world, local = env_rank_world()
    if world == 1:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist


def gather_rows(local, total: int, dst: int = 0, out=None):
    """Final gather of per-rank row blocks (shard_range order) to `dst`: returns the [total, ...] tensor on dst, None
    elsewhere.  Gather-to-root: every other rank sends its block ONCE, straight into its slot of dst's array (grouped
    point-to-point sends — with the nccl backend RCCL puts each on the peer's own xGMI link), so nothing is padded, nothing
    is concatenated afterwards and no rank but dst ever holds the whole result.  `out` (dst only): a preallocated
    [total, ...] tensor; when `local` already is dst's slot of it, dst's own block is not copied at all."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return local
    rank, world = dist.get_rank(), dist.get_world_size()
    sizes = shard_sizes(total, world)
    if local.shape[0] != sizes[rank]:
        raise ValueError("rank %d holds %d rows, expected %d" % (rank, local.shape[0], sizes[rank]))
    if rank != dst:
        if sizes[rank] > 0:
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, local.contiguous(), dst)]):
                req.wait()
        return None
    if out is None:
        out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    elif tuple(out.shape) != (total,) + tuple(local.shape[1:]) or not out.is_contiguous():
        raise ValueError("out must be a contiguous [total, ...] tensor")
    ops = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        if hi == lo:
            continue
        if r == dst:
            if out[lo:hi].data_ptr() != local.data_ptr():
                out[lo:hi].copy_(local)
        else:
            ops.append(dist.P2POp(dist.irecv, out[lo:hi], r))  # a block of whole rows: a contiguous view, received in place
    for req in (dist.batch_isend_irecv(ops) if ops else []):
        req.wait()
    return out


def max_over_ranks(value: float) -> float:
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return value
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"  # gloo reduces host tensors
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        dist.barrier()
