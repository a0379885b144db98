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
want = torch.arange(total, dtype=torch.float32)[:, None] * torch.tensor([[1.0, 10.0]])
if rank == 0:
    assert full is not None and torch.equal(full, want), full
    assert t == float(world)
else:
    assert full is None
# gather to another root, into a preallocated array whose slot the root computed its block in (no copy of its own rows)
buf = torch.full((total, 2), -1.0)
mine = buf[lo:hi]
mine.copy_(local)
full1 = D.gather_rows(mine if rank == 1 else local, total, dst=1, out=buf if rank == 1 else None)
if rank == 1:
    assert full1 is buf and torch.equal(buf, want), buf
else:
    assert full1 is None
# a rank with no rows at all (3 units over 2 ranks is fine; 1 unit leaves rank 1 empty)
lo1, hi1 = D.shard_range(1, rank, world)
one = D.gather_rows(torch.full((hi1 - lo1, 3), 7.0), 1)
if rank == 0:
    assert one.shape == (1, 3) and bool((one == 7.0).all())
    print("OK")
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


def test_gather_offsets_host_arithmetic():
    """smplpp_gather_offsets (include/smplpp_hip.h): the slot of every rank's block in the gathered array, shared by
    smplpp_gather's ragged leg and smplpp_gather_to_root.  Pure host code: runs without a GPU."""
    import ctypes as C

    from smplpp_amd import _lib

    L = _lib.load()
    for total, world, rf in [(1024, 8, 6890 * 3), (64, 8, 75), (11, 4, 2), (3, 4, 5), (0, 2, 3), (3163, 8, 75)]:
        sizes = D.shard_sizes(total, world)
        rows = (C.c_int64 * world)(*sizes)
        offs = (C.c_int64 * (world + 1))()
        assert L.smplpp_gather_offsets(rows, world, rf, offs) == 0
        want = np.concatenate([[0], np.cumsum(sizes)]) * rf
        assert list(offs) == want.tolist()
        for r in range(world):
            assert offs[r] == D.shard_range(total, r, world)[0] * rf
    bad = (C.c_int64 * 2)(3, -1)
    offs = (C.c_int64 * 3)()
    assert L.smplpp_gather_offsets(bad, 2, 4, offs) != 0
    huge = (C.c_int64 * 2)(2**40, 2**40)
    assert L.smplpp_gather_offsets(huge, 2, 2**30, offs) != 0  # overflow is refused, not wrapped


# ---- the launcher a program asked for `--gpus N` uses when nobody started its ranks (bench.py)
def test_launch_plan():
    assert D.launch_plan(1, {}) == "single"
    assert D.launch_plan(8, {}) == "spawn"
    assert D.launch_plan(8, {"WORLD_SIZE": "8", "RANK": "3"}) == "rank"
    assert D.launch_plan(1, {"WORLD_SIZE": "1"}) == "single"
    for gpus, ws in [(8, "1"), (2, "4"), (1, "2")]:  # someone else's WORLD_SIZE contradicts --gpus: refused, never measured as a smaller job
        with pytest.raises(SystemExit):
            D.launch_plan(gpus, {"WORLD_SIZE": ws})
    with pytest.raises(SystemExit):
        D.launch_plan(0, {})


LAUNCHED = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch
from smplpp_amd import dist as D
d = D.init_process_group("gloo")
rank, world, local = D.env_rank_world()
assert world == 2 and local == rank and os.environ["MASTER_ADDR"] == "127.0.0.1"
if {fail!r} and rank == 1:
    sys.exit(3)
full = D.gather_rows(torch.full((1, 2), float(rank)), 2)
m = D.max_over_ranks(float(rank))
print("not the line: rank", rank, file=sys.stderr)
if rank == 0:
    print('{{"n_gpus": %d, "sum": %g, "max": %g}}' % (world, float(full.sum()), m))
else:
    print("a rank other than 0 printed to stdout: must not be forwarded")
D.barrier()
d.destroy_process_group()
"""


def test_launch_ranks_two_gloo_ranks(tmp_path):
    import io

    script = tmp_path / "launched.py"
    script.write_text(LAUNCHED.format(root=ROOT, fail=False))
    out, err = io.StringIO(), io.StringIO()
    rc = D.launch_ranks([sys.executable, str(script)], 2, timeout=120, stdout=out, stderr=err)
    assert rc == 0, err.getvalue()
    lines = [ln for ln in out.getvalue().splitlines() if ln.strip() and not ln.startswith("[Gloo]")]  # (gloo's own chatter goes to stdout)
    assert lines == ['{"n_gpus": 2, "sum": 2, "max": 1}'], (lines, err.getvalue())


def test_launch_ranks_a_failing_rank_stops_the_job(tmp_path):
    import io

    script = tmp_path / "launched.py"
    script.write_text(LAUNCHED.format(root=ROOT, fail=True))  # rank 1 exits 3 while rank 0 waits in the gather
    out, err = io.StringIO(), io.StringIO()
    rc = D.launch_ranks([sys.executable, str(script)], 2, timeout=120, stdout=out, stderr=err)
    assert rc == 3 and "rank 1 exited with 3" in err.getvalue()
    assert "n_gpus" not in out.getvalue()


MEASURED = r"""
import os, sys, time
sys.path.insert(0, {root!r})
import numpy as np
import torch
from smplpp_amd import dist as D
d = D.init_process_group("gloo")
rank, world, local = D.env_rank_world()
assert os.environ.get("NCCL_DEBUG") == "WARN" and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
# rank 1 is the slow one: the job's time is ITS time, and rank 0's own clock must not contain the wait for it
sp = D.timed_region(lambda: time.sleep(0.6 if rank == 1 else 0.05), lambda: None)
# (generous upper bounds: a loaded host may deschedule a rank for a while; what matters is min << max)
assert sp["rank_of_max"] == 1 and 0.59 < sp["max"] < 3.0 and 0.04 < sp["min"] < 0.4, sp
assert len(sp["per_rank"]) == 2 and sp["per_rank"][0] == sp["min"]
c = D.count_ranks()
assert c == {{"collective_backend": "gloo", "ranks_counted_by_allreduce": 2}}, c
vals = D.gather_values(np.arange(3 if rank == 0 else 2, dtype=np.float64) + 10 * rank)
assert vals.tolist() == [0.0, 1.0, 2.0, 10.0, 11.0], vals
if rank == 0:
    print("MEASURED-OK")
D.barrier()
d.destroy_process_group()
"""


def test_timed_region_keeps_collectives_out_of_the_clock_and_counts_ranks_by_allreduce(tmp_path):
    """the closing barrier is outside every rank's clock (the fast rank's own time stays its own), the
    spread over ranks names the slow rank, and the rank count in the line comes from a collective, labelled with the backend that ran
    it ("gloo" here — never "rccl")."""
    import io

    script = tmp_path / "measured.py"
    script.write_text(MEASURED.format(root=ROOT))
    out, err = io.StringIO(), io.StringIO()
    rc = D.launch_ranks([sys.executable, str(script)], 2, timeout=120, stdout=out, stderr=err)
    assert rc == 0, err.getvalue()
    assert "MEASURED-OK" in out.getvalue()
    # not launched distributed: the helpers are the identity
    sp = D.timed_region(lambda: None, lambda: None)
    assert sp["rank_of_max"] == 0 and sp["max"] == sp["min"] and len(sp["per_rank"]) == 1
    assert D.count_ranks() == {"collective_backend": None, "ranks_counted_by_allreduce": 1}


NOISY_FAILURE = r"""
import os, sys, time
rank = int(os.environ["RANK"])
if rank == 1:
    for i in range(60):
        print("rank one diagnostic line %d" % i, file=sys.stderr)
    sys.exit(5)
for i in range(200):
    print("peer noise %d" % i, file=sys.stderr)
    time.sleep(0.01)
time.sleep(60)
"""


def test_launch_ranks_repeats_the_failed_ranks_last_lines(tmp_path):
    """(d) with N ranks writing at once the reason a rank died is buried in the peers' output: its last 40 stderr lines come again,
    under a header, once the job has stopped."""
    import io

    script = tmp_path / "noisy.py"
    script.write_text(NOISY_FAILURE)
    out, err = io.StringIO(), io.StringIO()
    rc = D.launch_ranks([sys.executable, str(script)], 2, timeout=120, stdout=out, stderr=err)
    text = err.getvalue()
    assert rc == 5
    head = "last 40 stderr lines of rank 1 (exit code 5)"
    assert head in text
    tail = text[text.index(head):]
    assert tail.count("[rank 1] rank one diagnostic line") == 40 and "[rank 1] rank one diagnostic line 59" in tail
    assert "line 19\n" not in tail and "peer noise" not in tail


PORT_RACE = r"""
import os, sys
marker = os.path.join({tmp!r}, "first_attempt_done")
if not os.path.exists(marker):
    if os.environ["RANK"] == "0":
        open(marker, "w").write(os.environ["MASTER_PORT"])
        print("RuntimeError: The server socket has failed to listen on any local network address. port: %s, EADDRINUSE: address already in use" % os.environ["MASTER_PORT"], file=sys.stderr)
        sys.exit(1)
    import time
    time.sleep(30)
print("second attempt on port", os.environ["MASTER_PORT"]) if os.environ["RANK"] == "0" else None
"""


def test_launch_ranks_retries_when_the_port_was_taken(tmp_path):
    """free_port() closes its socket before rank 0 binds the port: a launch that loses that race is started again."""
    import io

    script = tmp_path / "race.py"
    script.write_text(PORT_RACE.format(tmp=str(tmp_path)))
    out, err = io.StringIO(), io.StringIO()
    rc = D.launch_ranks([sys.executable, str(script)], 2, timeout=120, stdout=out, stderr=err)
    assert rc == 0, err.getvalue()
    assert "starting the ranks again on another port" in err.getvalue() and "second attempt on port" in out.getvalue()


def test_launch_plan_warns_under_a_profiler_preload(capsys):
    assert D.profiler_preloaded({"LD_PRELOAD": "/opt/rocm/lib/librocprofiler-sdk-tool.so"}) and D.profiler_preloaded({"ROCP_TOOL_LIBRARIES": "x"})
    assert not D.profiler_preloaded({"LD_PRELOAD": "/usr/lib/libjemalloc.so", "PATH": "/bin"})
    assert D.launch_plan(2, {"ROCP_TOOL_LIBRARIES": "x"}) == "spawn"
    assert "each of the 2 ranks" in capsys.readouterr().err
    assert D.launch_plan(1, {"ROCP_TOOL_LIBRARIES": "x"}) == "single" and capsys.readouterr().err == ""


def test_bench_starts_its_own_ranks_and_refuses_a_contradicting_world_size():
    """No GPU here: each rank bench.py starts fails loudly ("needs an MI355X"), and the parent — which must not have touched the
    GPU or torch.distributed itself — reports it and exits non-zero; WORLD_SIZE=1 with --gpus 2 is refused outright."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-ik"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert r.stderr.count("bench.py needs an MI355X") >= 1 and "launch_ranks: rank" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="1"), capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_launch_ranks_stopped_launcher_takes_its_ranks_along(tmp_path):
    """SIGTERM to the launcher (a driver that timed the job out): the ranks it started are terminated too — none is left holding a GPU."""
    import signal
    import time

    worker = tmp_path / "sleeper.py"
    worker.write_text("import os, sys, time\nopen(os.path.join(%r, 'pid%%s' %% os.environ['RANK']), 'w').write(str(os.getpid()))\ntime.sleep(120)\n" % str(tmp_path))
    parent = tmp_path / "parent.py"
    parent.write_text("import sys\nsys.path.insert(0, %r)\nfrom smplpp_amd import dist as D\nsys.exit(D.launch_ranks([sys.executable, %r], 2))\n" % (ROOT, str(worker)))
    p = subprocess.Popen([sys.executable, str(parent)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    deadline = time.time() + 60
    while time.time() < deadline and not all((tmp_path / ("pid%d" % r)).exists() and (tmp_path / ("pid%d" % r)).read_text() for r in range(2)):
        time.sleep(0.1)
    pids = [int((tmp_path / ("pid%d" % r)).read_text()) for r in range(2)]
    p.send_signal(signal.SIGTERM)
    p.wait(timeout=30)
    assert p.returncode != 0
    for pid in pids:
        for _ in range(100):
            try:
                os.kill(pid, 0)
            except ProcessLookupError:
                break
            time.sleep(0.1)
        else:
            os.kill(pid, signal.SIGKILL)  # (never leave the test's own children behind)
            raise AssertionError("rank process %d survived its launcher" % pid)
