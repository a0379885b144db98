"""Development aid: host time to ENQUEUE one FK step (python + ctypes + two kernel launches) against the GPU time per step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplpp_amd import model_io
from smplpp_amd.smpl import SMPL
n = 1024
s = SMPL(); s.setDevice("cuda:0"); s.init(model_io.synthetic_model())
b, t = model_io.synthetic_inputs(n)
bd, td = torch.from_numpy(b).cuda(), torch.from_numpy(t).cuda()
out = {"verts": torch.empty((n, 6890, 3), dtype=torch.float32, device="cuda")}
for _ in range(20): s.launch(bd, td, want=("verts",), out=out)
torch.cuda.synchronize()
for steps in (50, 400):
    t0 = time.perf_counter()
    for _ in range(steps): s.launch(bd, td, want=("verts",), out=out)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("steps %d: enqueue %.1f us/step, total %.1f us/step" % (steps, (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6))
