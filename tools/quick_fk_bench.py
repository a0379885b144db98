"""Quick FK timing (development aid; bench.py is the contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplpp_amd import model_io
from smplpp_amd.smpl import SMPL

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
s = SMPL(); s.setDevice("cuda:0"); s.init(model_io.synthetic_model())
b, t = model_io.synthetic_inputs(n)
bd, td = torch.from_numpy(b).cuda(), torch.from_numpy(t).cuda()
for _ in range(600 if n <= 4096 else 50): s.launch(bd, td, want=("verts",))  # past the chip's clock ramp (tools/fk_ramp.py)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps): s.launch(bd, td, want=("verts",))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps
bytes_alg = 19347120 + 83020 * n
print("n=%d  %.3f ms/step  %.3g evals/s  alg %.1f GB/s  %.1f TFLOP/s" % (n, ms, n / ms * 1e3, bytes_alg / ms / 1e6, 15.5e6 * n / ms / 1e9))
