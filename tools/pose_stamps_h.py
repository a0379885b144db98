"""Development aid: stamps of the in-kernel pose step of skin_kernel_h (variant built with -DSKINH_ABL=512 or 1536).
usage: SMPLPP_HIP_LIB=$PWD/ab/h1536.so python tools/pose_stamps_h.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplpp_amd import model_io, _lib
from smplpp_amd.smpl import SMPL

s = SMPL(); s.setDevice("cuda:0"); s.init(model_io.synthetic_model())
b, t = model_io.synthetic_inputs(1024)
bd, td = torch.from_numpy(b).cuda(), torch.from_numpy(t).cuda()
for _ in range(100): s.launch(bd, td, want=("verts",))
torch.cuda.synchronize()
L = _lib.load()
buf = (ctypes.c_ulonglong * (8 * 256))()
L.smplpp_debug_hslot_times.restype = ctypes.c_int
assert L.smplpp_debug_hslot_times(buf) == 0
T = np.array(buf, dtype=np.uint64).astype(np.int64)
names = ["entry", "staged", "inputs in regs", "rodrigues", "joints", "A chunks written", "barrier", "A read+barrier", "chain", "G' written", "exit"]
for rep in range(2):
    print("pass", rep)
    for w in range(4):
        st = T[256 + rep * 64 + w * 16: 256 + rep * 64 + w * 16 + 11]
        if st[0] == 0: continue
        print("  wave %d:" % w, " ".join("%s +%d" % (names[k], st[k] - st[k - 1]) for k in range(1, 11)), "| total", st[10] - st[0])
