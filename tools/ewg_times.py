"""Development aid: per-workgroup start / end / cycles of skin_kernel_e (variant built with -DSKINE_ABL=512 [+ other bits]).
usage: SMPLPP_HIP_LIB=$PWD/ab/e512.so python tools/ewg_times.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplpp_amd import model_io, _lib
from smplpp_amd.smpl import SMPL

s = SMPL(); s.setDevice("cuda:0"); s.init(model_io.synthetic_model())
b, t = model_io.synthetic_inputs(1024)
bd, td = torch.from_numpy(b).cuda(), torch.from_numpy(t).cuda()
for _ in range(600): s.launch(bd, td, want=("verts",))
torch.cuda.synchronize()
L = _lib.load()
wb = (ctypes.c_ulonglong * (256 * 8))()
L.smplpp_debug_ewg_times.restype = ctypes.c_int
assert L.smplpp_debug_ewg_times(wb) == 0
W = np.array(wb, dtype=np.uint64).reshape(256, 8).astype(np.int64)
st, en, cyc, items = W[:, 0], W[:, 1], W[:, 2], W[:, 3]
t00 = st.min()
print("all workgroups: first start 0, last start %.2f us, first end %.2f us, last end %.2f us" % ((st.max() - t00) / 100.0, (en.min() - t00) / 100.0, (en.max() - t00) / 100.0))
dur = (en - st) / 100.0
pro, drn, mid, body = W[:, 4], W[:, 5], W[:, 6], W[:, 7]
print("cycles: prologue mean %.0f (min %d max %d); final drain mean %.0f; run change (drain + set-up; %d workgroups have one) mean %.0f; items mean %.0f" % (
    pro.mean(), pro.min(), pro.max(), drn.mean(), (mid > 0).sum(), mid[mid > 0].mean() if (mid > 0).any() else 0, ((body - mid) / items).mean()))
for it in sorted(set(items.tolist())):
    m = items == it
    print("  %d items: %d workgroups, duration %.2f .. %.2f us (mean %.2f), cycles mean %.0f (%.0f per item), clock %.0f MHz" % (
        it, m.sum(), dur[m].min(), dur[m].max(), dur[m].mean(), cyc[m].mean(), cyc[m].mean() / it, (cyc[m] / np.maximum(en[m] - st[m], 1)).mean() * 100))
