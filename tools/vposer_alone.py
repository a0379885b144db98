"""Development aid: the decoder kernel alone (value + Jacobian), device-resident operands, HIP-event time per launch at several
frame counts.  usage: [SMPLPP_HIP_LIB=...] python tools/vposer_alone.py [frames ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplpp_amd import _lib
from smplpp_amd._lib import DEVICE, check
from smplpp_amd.ik import VPoserDecoder
vp = VPoserDecoder(VPoserDecoder.synthetic_params(seed=3), device=0)
L = _lib.load()
for n in ([int(a) for a in sys.argv[1:]] or [8, 64, 256, 512]):
    z = torch.from_numpy(np.random.default_rng(0).normal(0, 0.3, (n, 32)).astype(np.float32)).cuda()
    out = torch.empty((n, 63), dtype=torch.float32, device="cuda"); jac = torch.empty((n, 63, 32), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: C.c_void_p(t.data_ptr())
    def go(k):
        for _ in range(k): check(L.smplpp_vposer_forward_at(vp._h, n, 0, p(z), p(out), p(jac), DEVICE, C.c_void_p(st)))
    go(50); torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); go(200); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 200 * 1e3)
    print("decoder alone, %4d frames: %.1f us per launch" % (n, best))
