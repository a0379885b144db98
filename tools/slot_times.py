"""Development aid: per-slot s_memtime deltas of one wavefront of skin_kernel_b (variant built with -DSKINB_ABL=256).
usage: SMPLPP_HIP_LIB=$PWD/ab/slots.so python tools/slot_times.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplpp_amd import model_io, _lib
from smplpp_amd.smpl import SMPL

s = SMPL(); s.setDevice("cuda:0"); s.init(model_io.synthetic_model())
b, t = model_io.synthetic_inputs(1024)
bd, td = torch.from_numpy(b).cuda(), torch.from_numpy(t).cuda()
for _ in range(5): s.launch(bd, td, want=("verts",))
torch.cuda.synchronize()
L = _lib.load()
buf = (ctypes.c_ulonglong * (8 * 256))()
L.smplpp_debug_slot_times.restype = ctypes.c_int
assert L.smplpp_debug_slot_times(buf) == 0
T = np.array(buf, dtype=np.uint64).reshape(8, 256)[:, :252].astype(np.int64)
for it in (1, 2, 3, 4):
    d = np.diff(T[it])  # 251 deltas
    total = T[it + 1][0] - T[it][0] if it + 1 < 7 else 0
    print("item %d: slots sum %d ticks (next-item start to start %d), median slot %d, mean %.1f" % (it, d.sum(), total, np.median(d), d.mean()))
d = np.diff(T[2])
M = np.arange(251) % 18
print("mean ticks by slot-in-k-step M:", " ".join("%d:%.0f" % (m, d[M == m].mean()) for m in range(18)))
ks = np.arange(251) // 18
print("ticks per k-step:", " ".join("%d" % d[ks == k].sum() for k in range(14)))
big = np.argsort(-d)[:12]
print("slowest slots (S, ticks):", [(int(i), int(d[i])) for i in big])
