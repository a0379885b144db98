"""Development aid: phase stamps of pose_kernel (variant built with -DPOSE_STAMP). usage: SMPLPP_HIP_LIB=$PWD/ab/pst.so python tools/pose_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplpp_amd import model_io, _lib
from smplpp_amd.smpl import SMPL
s = SMPL(); s.setDevice("cuda:0"); s.init(model_io.synthetic_model())
b, t = model_io.synthetic_inputs(1024)
bd, td = torch.from_numpy(b).cuda(), torch.from_numpy(t).cuda()
for _ in range(50): s.launch(bd, td, want=("verts",))
torch.cuda.synchronize()
L = _lib.load()
buf = (ctypes.c_ulonglong * 16)()
L.smplpp_debug_pose_stamps.restype = ctypes.c_int
assert L.smplpp_debug_pose_stamps(buf) == 0
tc = np.array(buf, dtype=np.uint64).astype(np.int64)
t = tc[:7]
names = ["loads + Rodrigues (thread 0)", "barrier 1", "phase 1 (chain beside it) + barrier 2", "phase 2 (thread 0: A2h)", "barrier 3 (generic trees only)", "phase 3 stores"]
for i in range(6): print("%-34s %6d cycles" % (names[i], t[i + 1] - t[i]))
print("total %d cycles" % (t[6] - t[0]))
print("chain wavefront (fast path): barrier 1 -> start %d, operand prefetch %d, levels %d, -> barrier 2 passed (thread 0) %d" % (
    tc[8] - tc[2], tc[9] - tc[8], tc[10] - tc[9], tc[3] - tc[10]))
