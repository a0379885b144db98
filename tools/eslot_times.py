"""Development aid: per-slot cycle-counter deltas of one wavefront of skin_kernel_e (variant built with -DSKINE_ABL=256).
usage: SMPLPP_HIP_LIB=$PWD/ab/e256.so python tools/eslot_times.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplpp_amd import model_io, _lib
from smplpp_amd.smpl import SMPL

s = SMPL(); s.setDevice("cuda:0"); s.init(model_io.synthetic_model())
b, t = model_io.synthetic_inputs(1024)
bd, td = torch.from_numpy(b).cuda(), torch.from_numpy(t).cuda()
for _ in range(300): s.launch(bd, td, want=("verts",))
torch.cuda.synchronize()
L = _lib.load()
buf = (ctypes.c_ulonglong * (8 * 256))()
L.smplpp_debug_eslot_times.restype = ctypes.c_int
assert L.smplpp_debug_eslot_times(buf) == 0
T = np.array(buf, dtype=np.uint64).reshape(8, 256)[:, :253].astype(np.int64)
for it in range(0, 6):
    d = np.diff(T[it])  # 252 deltas (the last: slot 251 -> end of the item)
    gap = T[it + 1][0] - T[it][252] if it + 1 < 8 and T[it + 1][0] > 0 else 0
    print("item %d: %d cycles (%.1f per slot), median slot %d; to the next item's first slot %d" % (it, d.sum(), d.mean(), np.median(d), gap))
for it in (0, 2):
    d = np.diff(T[it])
    M = np.arange(252) % 18
    print("item %d mean cycles by slot-in-k-step M:" % it, " ".join("%d:%.0f" % (m, d[M == m].mean()) for m in range(18)))
    ks = np.arange(252) // 18
    print("item %d cycles per k-step:" % it, " ".join("%d" % d[ks == k].sum() for k in range(14)))
    big = np.argsort(-d)[:14]
    print("item %d slowest slots (S, cycles):" % it, [(int(i), int(d[i])) for i in big])
d = np.diff(T[2])
print("item 2, all slots:")
for k in range(14):
    print("  ks %2d: %s" % (k, " ".join("%3d" % x for x in d[k * 18:(k + 1) * 18])))
