"""smplpp_gather (include/smplpp_hip.h): the final gather of the multi-GPU split through the C ABI, on a communicator the
HOST created with RCCL.  One GPU box = one rank: the equal-blocks leg (one ncclAllGather) end to end; the ragged leg needs
more than one device and is covered by construction (dist.shard_sizes order) in tests/test_dist_cpu.py."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rccl():
    for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
        try:
            return C.CDLL(name, mode=C.RTLD_GLOBAL)
        except OSError:
            continue
    import os

    return C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=C.RTLD_GLOBAL)


def _unique_id_comm(rccl):
    """A communicator the way a multi-process C++ host makes one: ncclGetUniqueId on rank 0, ncclCommInitRank on every rank (here:
    world 1), the id passed BY VALUE (ncclUniqueId is a 128-byte struct)."""
    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    return comm


def test_send_recv_binding_executes_on_a_one_rank_communicator():
    """the grouped ncclSend / ncclRecv leg of smplpp_gather_to_root has no peer to talk to on a one-GPU box;
    smplpp_gather_selfcheck drives the same bound function pointers, datatype constant and group calls as a self-exchange, on a
    communicator from ncclCommInitRank (the multi-process way), and the floats must arrive."""
    from smplpp_amd import _lib

    L = _lib.load()
    rccl = _rccl()
    torch.cuda.set_device(0)
    comm = _unique_id_comm(rccl)
    try:
        n = 6890 * 3 * 2 + 5
        send = torch.arange(n, dtype=torch.float32, device="cuda:0") * 0.25 - 7.0
        recv = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda:0")
        rc = L.smplpp_gather_selfcheck(comm, 0, send.data_ptr(), recv.data_ptr(), n, None)
        assert rc == 0, L.smplpp_last_error()
        torch.cuda.synchronize()
        assert torch.equal(send, recv)
        # on a side stream as a host would issue it, and the argument checks
        st = torch.cuda.Stream()
        recv.fill_(-1.0)
        torch.cuda.synchronize()
        assert L.smplpp_gather_selfcheck(comm, 0, send.data_ptr(), recv.data_ptr(), 1000, C.c_void_p(st.cuda_stream)) == 0, L.smplpp_last_error()
        st.synchronize()
        assert torch.equal(send[:1000], recv[:1000]) and bool((recv[1000:] == -1.0).all())
        assert L.smplpp_gather_selfcheck(comm, 0, send.data_ptr(), send.data_ptr(), 10, None) != 0
        assert L.smplpp_gather_selfcheck(None, 0, send.data_ptr(), recv.data_ptr(), 10, None) != 0
        # and the gather itself on this communicator (one rank: root's block by a device copy)
        per = (C.c_int64 * 1)(3)
        out = torch.zeros(3 * 75, dtype=torch.float32, device="cuda:0")
        assert L.smplpp_gather_to_root(comm, send.data_ptr(), out.data_ptr(), per, 1, 0, 0, 75, None) == 0, L.smplpp_last_error()
        torch.cuda.synchronize()
        assert torch.equal(out, send[:225])
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_gather_one_rank_roundtrip():
    from smplpp_amd import _lib

    L = _lib.load()
    rccl = _rccl()
    comm = C.c_void_p()
    dev = (C.c_int * 1)(0)
    rccl.ncclCommInitAll.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
    assert rccl.ncclCommInitAll(C.byref(comm), 1, dev) == 0
    try:
        rows, rf = 37, 75
        send = torch.arange(rows * rf, dtype=torch.float32, device="cuda:0").reshape(rows, rf) * 0.5
        recv = torch.full((rows, rf), -1.0, dtype=torch.float32, device="cuda:0")
        per = (C.c_int64 * 1)(rows)
        rc = L.smplpp_gather(comm, send.data_ptr(), recv.data_ptr(), per, 1, 0, rf, None)
        assert rc == 0, L.smplpp_last_error()
        torch.cuda.synchronize()
        assert torch.equal(send, recv)
        # gather-to-root on the one-rank communicator: root's own block by a device copy; in place when send is its slot
        recv2 = torch.full((rows, rf), -2.0, dtype=torch.float32, device="cuda:0")
        assert L.smplpp_gather_to_root(comm, send.data_ptr(), recv2.data_ptr(), per, 1, 0, 0, rf, None) == 0, L.smplpp_last_error()
        torch.cuda.synchronize()
        assert torch.equal(send, recv2)
        assert L.smplpp_gather_to_root(comm, recv2.data_ptr(), recv2.data_ptr(), per, 1, 0, 0, rf, None) == 0
        assert L.smplpp_gather_to_root(comm, send.data_ptr(), None, per, 1, 0, 0, rf, None) != 0
        # argument checks
        assert L.smplpp_gather(None, send.data_ptr(), recv.data_ptr(), per, 1, 0, rf, None) != 0
        assert L.smplpp_gather(comm, send.data_ptr(), recv.data_ptr(), per, 1, 1, rf, None) != 0
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


NCCL_ONE_RANK = r"""
import os, sys, time
sys.path.insert(0, {root!r})
import numpy as np
import torch
import torch.distributed as tdist
from smplpp_amd import dist as D
torch.cuda.set_device(0)
tdist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:{port}", rank=0, world_size=1)
c = D.count_ranks()
assert c == {{"collective_backend": "nccl", "ranks_counted_by_allreduce": 1}}, c
sp = D.timed_region(lambda: time.sleep(0.01), torch.cuda.synchronize)
assert sp["rank_of_max"] == 0 and len(sp["per_rank"]) == 1 and 0.009 < sp["max"] < 0.5, sp
assert D.max_over_ranks(3.5) == 3.5
rows = torch.arange(12, dtype=torch.float32, device="cuda").reshape(4, 3)
full = D.gather_rows(rows, 4)
assert full is not None and torch.equal(full, rows)
vals = D.gather_values(np.array([1.0, 2.0, 5.0]))
assert vals.tolist() == [1.0, 2.0, 5.0]
D.barrier()
tdist.destroy_process_group()
print("NCCL-ONE-RANK-OK")
"""


def test_measurement_helpers_under_the_nccl_backend_one_rank(tmp_path):
    """dist.count_ranks / timed_region / max_over_ranks / gather_rows / gather_values with torch.distributed's nccl backend (= RCCL):
    the device-tensor branches of the helpers the N > 1 bench line is made of.  RCCL refuses a second rank on the same device
    (profiles/r05_rccl_same_device_probe.txt), so one rank is what a one-GPU box can run; the two-rank control flow runs under gloo
    (tests/test_dist_cpu.py)."""
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "nccl_one_rank.py"
    script.write_text(NCCL_ONE_RANK.format(root=root, port=port))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "NCCL-ONE-RANK-OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
