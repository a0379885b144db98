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
