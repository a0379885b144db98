"""Multi-GPU: frames (FK) or whole sequences/restarts (IK) are independent, so a job shards embarrassingly — one
process per GPU, contiguous blocks per rank, NO collective inside the compute — and the only exchange is the final
gather of results (RCCL over xGMI: `torch.distributed` backend "nccl" is RCCL on ROCm; "gloo" on CPU for tests).

The reference has no counterpart (single process, device index 0: node/node.cpp:372); SURVEY.md §8(e) is the spec.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
import time
from typing import List, Optional, Sequence, Tuple

import numpy as np


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment; (0, 1, 0) when not launched distributed."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def launch_plan(gpus: int, environ=None) -> str:
    """What a program asked for `gpus` ranks has to do, from its environment alone (no GPU call): "single" (one rank, not
    launched distributed), "rank" (it IS one rank of a launched job: torchrun or launch_ranks set RANK / WORLD_SIZE), or
    "spawn" (gpus > 1 and nobody launched the ranks: the program starts them itself, launch_ranks).  A WORLD_SIZE set by
    someone else that contradicts `gpus` is refused — never silently measured as a smaller job."""
    environ = os.environ if environ is None else environ
    if gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    ws = environ.get("WORLD_SIZE")
    if ws is None:
        if gpus > 1 and profiler_preloaded(environ):
            # the ranks are FRESH processes (allowed: nothing here re-execs a process that touched the GPU), but each inherits the
            # profiler's preload and writes its own trace — say so instead of leaving N interleaved outputs unexplained
            sys.stderr.write("launch_plan: a profiler preload is active (LD_PRELOAD / ROCP*): each of the %d ranks this program starts "
                             "will be profiled on its own; to profile ONE rank, run it under torch.distributed.run instead\n" % gpus)
        return "single" if gpus == 1 else "spawn"
    if int(ws) != gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%s in the environment: launch %d ranks, or unset WORLD_SIZE and let the "
                         "program start its own" % (gpus, ws, gpus))
    return "single" if gpus == 1 else "rank"


def profiler_preloaded(environ=None) -> bool:
    """rocprofv3 and friends work by preloading a tool library into the program (LD_PRELOAD, ROCP_TOOL_LIBRARIES / ROCPROFILER_*)."""
    environ = os.environ if environ is None else environ
    pre = environ.get("LD_PRELOAD", "")
    return ("rocprof" in pre) or any(k.startswith(("ROCP_", "ROCPROFILER_", "ROCPROF_")) for k in environ)


def free_port() -> int:
    """A port that was free a moment ago.  The socket is closed before the ranks bind it (torch's TCPStore cannot adopt a bound
    socket), so two jobs started at the same instant can still collide: launch_ranks retries the whole launch on EADDRINUSE."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _CountingWriter:
    """forwards to a stream and counts what went through (launch_ranks: has rank 0 said anything yet?)"""

    def __init__(self, dst):
        self.dst, self.count = dst, 0

    def write(self, text):
        self.count += len(text)
        return self.dst.write(text)

    def flush(self):
        return self.dst.flush()


RENDEZVOUS_SECONDS = 30.0  # a bind failure later than this is not the rendezvous
TAIL_LINES = 40  # of a failed rank's stderr, repeated under a header once the job has stopped
PORT_RETRIES = 3


def launch_ranks(argv: Sequence[str], world: int, extra_env: Optional[dict] = None, timeout: Optional[float] = None,
                 stdout=None, stderr=None) -> int:
    """Start `world` FRESH child processes of `argv` — one per rank, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT in their environment, rendezvous on 127.0.0.1 — wait for them, forward rank 0's stdout (every rank's stderr
    goes to ours as it comes) and return 0 only if every rank exited 0.  The caller must not have touched the GPU: children
    are new processes (fork + exec of the interpreter), never a re-exec of this one.  When a rank fails the others are
    terminated (their own process handles only), so a dead peer cannot leave the job hanging in a barrier; the first
    non-zero exit code is returned and the failed rank's last TAIL_LINES stderr lines are repeated under a header (with N
    ranks writing at once, the reason is otherwise buried in the peers' shutdown noise).
    Environment of the ranks, beyond the rendezvous: HSA_ENABLE_IPC_MODE_LEGACY=0 unless the caller set it (this pool's host
    driver only supports dmabuf IPC: without it RCCL's buffer exchange between processes fails with `hipIpcGetMemHandle:
    invalid argument`); NCCL_DEBUG=WARN unless set (RCCL then names the failing call instead of "unhandled system error").
    A rendezvous port that was taken between free_port() and rank 0's bind (EADDRINUSE) restarts the launch on a new port."""
    if world < 1:
        raise ValueError("world must be >= 1")
    stdout = sys.stdout if stdout is None else stdout
    stderr = sys.stderr if stderr is None else stderr
    rc = 0
    t_all = time.monotonic()
    for attempt in range(PORT_RETRIES):
        left = None if timeout is None else max(1.0, timeout - (time.monotonic() - t_all))  # `timeout` covers ALL attempts
        counted = _CountingWriter(stdout)
        t0 = time.monotonic()
        rc, tails = _launch_once(argv, world, extra_env, left, counted, stderr)
        if rc == 0:
            return 0
        # only a failure of the RENDEZVOUS is retried: the bind error in rank 0's tail, within seconds of the start, and nothing of
        # rank 0's stdout forwarded yet (a bind error minutes into a job is someone else's socket; a second run would repeat output)
        text = "".join(tails.get(0, ()))
        early = time.monotonic() - t0 < RENDEZVOUS_SECONDS and counted.count == 0
        bind = ("EADDRINUSE" in text or "Address already in use" in text or "address already in use" in text)
        if attempt + 1 < PORT_RETRIES and early and bind:
            stderr.write("launch_ranks: the rendezvous port was taken by someone else; starting the ranks again on another port\n")
            continue
        return rc
    return rc


def _launch_once(argv, world, extra_env, timeout, stdout, stderr):
    import collections

    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("NCCL_DEBUG", "WARN")
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen(list(argv), env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
    tails = {r: collections.deque(maxlen=TAIL_LINES) for r in range(world)}

    def pump(src, dst, keep=None):
        for ln in src:
            if keep is not None:
                keep.append(ln)
            dst.write(ln)
            dst.flush()

    threads = [threading.Thread(target=pump, args=(procs[0].stdout, stdout), daemon=True)]
    threads += [threading.Thread(target=pump, args=(p.stderr, stderr, tails[r]), daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    t0 = time.monotonic()
    live = set(range(world))
    failed: List[int] = []

    def stop_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    # a launcher that is told to stop (SIGTERM from whoever timed it, Ctrl-C) takes its ranks with it: ranks left behind would keep
    # their GPUs busy after the job is gone
    import signal

    def on_term(signum, frame):
        raise KeyboardInterrupt

    old_term = None
    if threading.current_thread() is threading.main_thread():
        old_term = signal.signal(signal.SIGTERM, on_term)
    try:
        rc = _wait_ranks(procs, live, world, timeout, t0, stderr, failed)
    except BaseException:
        stop_all()
        raise
    finally:
        if old_term is not None:
            signal.signal(signal.SIGTERM, old_term)
    for t in threads:
        t.join(timeout=5)
    for r in failed:
        stderr.write("launch_ranks: ---- last %d stderr lines of rank %d (exit code %s) ----\n" % (len(tails[r]), r, procs[r].returncode))
        for ln in tails[r]:
            stderr.write("  [rank %d] %s" % (r, ln if ln.endswith("\n") else ln + "\n"))
        stderr.flush()
    return rc, tails


def _wait_ranks(procs, live, world, timeout, t0, stderr, failed=None) -> int:
    rc = 0
    while live:
        for r in sorted(live):
            c = procs[r].poll()
            if c is None:
                continue
            live.discard(r)
            if c != 0 and rc == 0:
                rc = c
                if failed is not None:
                    failed.append(r)
                stderr.write("launch_ranks: rank %d exited with %d; stopping the other ranks\n" % (r, c))
        timed_out = timeout is not None and time.monotonic() - t0 > timeout
        if (rc != 0 or timed_out) and live:
            if timed_out and rc == 0:
                rc = 124
                stderr.write("launch_ranks: timeout after %.0f s\n" % timeout)
            for r in live:
                procs[r].terminate()
            for r in sorted(live):
                try:
                    procs[r].wait(timeout=10)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            live.clear()
        if live:
            time.sleep(0.05)
    return rc


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
    t = torch.tensor([value], dtype=torch.float64, device=_collective_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def _collective_device():
    import torch.distributed as dist

    return "cuda" if dist.get_backend() == "nccl" else "cpu"  # gloo reduces host tensors


def spread_over_ranks(value: float) -> dict:
    """{"max", "min", "rank_of_max", "per_rank"} of one number per rank (an all-gather of 8 bytes), so a slow rank is visible in
    the line and not only the maximum it causes.  Not launched distributed: the number itself."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return {"max": float(value), "min": float(value), "rank_of_max": 0, "per_rank": [float(value)]}
    t = torch.tensor([value], dtype=torch.float64, device=_collective_device())
    got = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(got, t)
    per = [float(g.item()) for g in got]
    return {"max": max(per), "min": min(per), "rank_of_max": int(np.argmax(per)), "per_rank": per}


def gather_values(values) -> np.ndarray:
    """Every rank's 1-D array of per-unit results (ragged: shard sizes differ), concatenated in rank order on every rank — for the
    statistics a line reports over the WHOLE job (medians, counts).  A few kilobytes, outside every timed region."""
    import torch.distributed as dist

    values = np.asarray(values)
    if not (dist.is_available() and dist.is_initialized()):
        return values
    got = [None] * dist.get_world_size()
    dist.all_gather_object(got, values)
    return np.concatenate([np.asarray(g).reshape(-1) for g in got]) if got else values


def timed_region(work, sync) -> dict:
    """The contract's bracket with NO collective inside the clock: sync + barrier, start the clock, `work()`, `sync()`, stop THIS
    rank's clock, then the closing barrier and the spread over ranks (its "max" is the job's time).  A barrier inside the clock
    would put one RCCL round trip (tens of microseconds at eight ranks) into a region that is one millisecond long at --steps 20,
    on a path whose data plane has no communication at all; the maximum over ranks already is "everyone has finished"."""
    sync()
    barrier()
    t0 = time.perf_counter()
    work()
    sync()
    dt = time.perf_counter() - t0
    barrier()
    return spread_over_ranks(dt)


def count_ranks() -> dict:
    """Evidence that the COLLECTIVE library saw every rank: an all-reduce (sum) of a one per rank, on the device under backend
    "nccl" (= RCCL: the sum travels over xGMI), on the host under gloo.  torch.distributed's own world size is configuration, not
    evidence.  {"collective_backend", "ranks_counted_by_allreduce"}."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return {"collective_backend": None, "ranks_counted_by_allreduce": 1}
    t = torch.ones(1, dtype=torch.float32, device=_collective_device())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return {"collective_backend": str(dist.get_backend()), "ranks_counted_by_allreduce": int(round(float(t.item())))}
