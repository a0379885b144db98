"""Development aid: FK step time against the length of the run (is a 200-step region a burst, or a chip still ramping up?)."""
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
def run(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): s.launch(bd, td, want=("verts",), out=out)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e6
for _ in range(5): s.launch(bd, td, want=("verts",), out=out)
time.sleep(1.0)
print("after 1 s idle: 20 steps %.1f us/step" % run(20))
time.sleep(1.0)
for steps in (20, 200, 2000, 20000, 2000, 200, 20):
    print("%6d steps back to back: %.1f us/step" % (steps, run(steps)))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
time.sleep(1.0)
ev[0].record()
for i in range(40):
    for _ in range(100): s.launch(bd, td, want=("verts",), out=out)
    ev[i + 1].record()
torch.cuda.synchronize()
print("4000 steps after 1 s idle, per block of 100 (us/step):", " ".join("%.1f" % (ev[i].elapsed_time(ev[i + 1]) * 10) for i in range(40)))
