"""Development aid: phase stamps of vposer_jac_kernel (variant built with -DVPJ_STAMP). usage: SMPLPP_HIP_LIB=$PWD/ab/vpj.so python tools/vpj_stamps.py [frames] [value]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplpp_amd import _lib
from smplpp_amd.ik import VPoserDecoder
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
vp = VPoserDecoder(VPoserDecoder.synthetic_params(seed=3), device=0)
z = np.random.default_rng(0).normal(0, 0.3, (n, 32)).astype(np.float32)
L = _lib.load()
if len(sys.argv) > 2 and sys.argv[2] == "value":  # the value-only instantiation (the capture loops' main stream)
    out = np.zeros((n, 63), np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    L.smplpp_debug_vposer_value.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, fp, fp]
    for _ in range(5): assert L.smplpp_debug_vposer_value(vp._h, n, 0, z.ctypes.data_as(fp), out.ctypes.data_as(fp)) == 0
else:
    for _ in range(5): vp.forward(z, want_jac=True)
buf = (ctypes.c_ulonglong * 16)()
L.smplpp_debug_vpj_stamps.restype = ctypes.c_int
assert L.smplpp_debug_vpj_stamps(buf) == 0
tt = np.array(buf, dtype=np.uint64).astype(np.int64)
t = tt[:7]
rt = (tt[14] - tt[8]) * 0.01  # s_memrealtime: 100 MHz
print("workgroup 0: %.1f us, %d ticks of s_memtime -> %.2f GHz" % (rt, t[6] - t[0], (t[6] - t[0]) / (rt * 1e3)))
names = ["layer 0 + fragments", "layer 1 loop", "activations + D2f", "layer 2", "rotation tail", "chain rule"]
for i in range(6): print("%-20s %7d cycles" % (names[i], t[i + 1] - t[i]))
print("total %d cycles" % (t[6] - t[0]))
