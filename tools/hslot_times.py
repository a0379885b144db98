"""Development aid: per-slot s_memtime deltas of one wavefront of skin_kernel_h (variant built with -DSKINH_ABL=256).
usage: SMPLPP_HIP_LIB=$PWD/ab/h256.so python tools/hslot_times.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplpp_amd import model_io, _lib
from smplpp_amd.smpl import SMPL

s = SMPL(); s.setDevice("cuda:0"); s.init(model_io.synthetic_model())
b, t = model_io.synthetic_inputs(1024)
bd, td = torch.from_numpy(b).cuda(), torch.from_numpy(t).cuda()
for _ in range(20): s.launch(bd, td, want=("verts",))
torch.cuda.synchronize()
L = _lib.load()
buf = (ctypes.c_ulonglong * (8 * 256))()
L.smplpp_debug_hslot_times.restype = ctypes.c_int
assert L.smplpp_debug_hslot_times(buf) == 0
T = np.array(buf, dtype=np.uint64).reshape(8, 256)[:, :186].astype(np.int64)
for it in range(0, 6):
    d = np.diff(T[it])
    nxt = T[it + 1][0] - T[it][0]
    print("item %d: gemm %d ticks, blend %d ticks, start-to-start %d" % (it, T[it][126] - T[it][0], T[it][185] - T[it][126], nxt))
d = np.diff(T[2])
print("GEMM ticks per k-step:", " ".join("%d" % (T[2][9 * (k + 1)] - T[2][9 * k]) for k in range(13)), "| last:", T[2][126] - T[2][117])
print("GEMM mean ticks by slot M:", " ".join("%d:%.0f" % (m, np.mean([T[2][9 * k + m + 1] - T[2][9 * k + m] for k in range(13)])) for m in range(8)))
print("blend ticks per entry:", " ".join("%d" % (T[2][126 + 5 * (e + 1)] - T[2][126 + 5 * e]) for e in range(11)))
print("blend mean ticks by slot B:", " ".join("%d:%.0f" % (m, np.mean([T[2][126 + 5 * e + m + 1] - T[2][126 + 5 * e + m] for e in range(11)])) for m in range(4)))
