"""Development aid: phase timestamps of one workgroup of skin_kernel_h (variant built with -DSKINH_ABL=512 [+ other bits]).
usage: SMPLPP_HIP_LIB=$PWD/ab/h512.so python tools/hphase_times.py"""
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
L.smplpp_debug_hslot_times.restype = ctypes.c_int
assert L.smplpp_debug_hslot_times(buf) == 0
T = np.array(buf, dtype=np.uint64).astype(np.int64)
items = int(T[68])
t0, r0, t1, r1 = T[66], T[67], T[64], T[65]
clk = (t1 - t0) / max(r1 - r0, 1) * 100.0
print("workgroup: %d items, %d cycles, %.2f us, clock %.0f MHz" % (items, t1 - t0, (r1 - r0) / 100.0, clk))
print("prologue (start -> first item): %d cycles; start -> entry barrier %d, -> loads issued %d, -> slot 0 landed %d; first item slots 0..7 start at +%s" % (T[0] - t0, T[70] - t0, T[71] - t0, T[72] - t0, [int(T[80 + k] - T[0]) for k in range(8)]))
for it in range(items):
    g0, b0 = T[it * 8 + 0], T[it * 8 + 2]
    nxt = T[(it + 1) * 8] if it + 1 < items else t1
    print("item %d: gemm %d cycles (%.0f per MFMA), blend+rest %d cycles (%.0f per MFMA)" % (it, b0 - g0, (b0 - g0) / 126.0, nxt - b0, (nxt - b0) / 60.0))

wb = (ctypes.c_ulonglong * (256 * 4))()
L.smplpp_debug_hwg_times.restype = ctypes.c_int
assert L.smplpp_debug_hwg_times(wb) == 0
W = np.array(wb, dtype=np.uint64).reshape(256, 4).astype(np.int64)
st, en, cyc, meta = W[:, 0], W[:, 1], W[:, 2], W[:, 3]
items = meta & 0xffffffff
xcc = (meta >> 32) & 0xf
t00 = st.min()
print("all workgroups: first start 0, last start %.2f us, first end %.2f us, last end %.2f us" % ((st.max() - t00) / 100.0, (en.min() - t00) / 100.0, (en.max() - t00) / 100.0))
dur = (en - st) / 100.0
for it in sorted(set(items.tolist())):
    m = items == it
    print("  %d items: %d workgroups, duration %.2f .. %.2f us (mean %.2f), cycles mean %.0f, clock %.0f MHz" % (it, m.sum(), dur[m].min(), dur[m].max(), dur[m].mean(), cyc[m].mean(), (cyc[m] / np.maximum(en[m] - st[m], 1)).mean() * 100))
print("  blockIdx & 7 -> XCC_ID:", [sorted(set(xcc[np.arange(256) % 8 == x].tolist())) for x in range(8)])
for x in range(8):
    m = (np.arange(256) % 8 == x)
    print("  XCD %d: items %s, duration mean %.2f max %.2f us, end (rel. first start) mean %.2f max %.2f, clock %.0f MHz" % (
        x, sorted(set(items[m].tolist())), dur[m].mean(), dur[m].max(), ((en[m] - t00) / 100.0).mean(), ((en[m] - t00) / 100.0).max(),
        (cyc[m] / np.maximum(en[m] - st[m], 1)).mean() * 100))
for it in sorted(set(items.tolist())):
    for x in range(8):
        m = (items == it) & (np.arange(256) % 8 == x)
        if m.sum(): print("  items %d XCD %d: n %d dur %.2f..%.2f" % (it, x, m.sum(), dur[m].min(), dur[m].max()))
