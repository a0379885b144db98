"""Development aid: what the rest-shape output costs the FK step at the IK loops' batch sizes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplpp_amd import model_io
from smplpp_amd.smpl import SMPL
s = SMPL(); s.setDevice("cuda:0"); s.init(model_io.synthetic_model())
for n in (64, 256, 512, 1024):
    b, t = model_io.synthetic_inputs(n)
    bd, td = torch.from_numpy(b).cuda(), torch.from_numpy(t).cuda()
    for want in (("verts",), ("verts", "rest")):
        out = {k: torch.empty((n, 6890, 3), dtype=torch.float32, device="cuda") for k in want}
        for _ in range(600): s.launch(bd, td, want=want, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(1000): s.launch(bd, td, want=want, out=out)
        torch.cuda.synchronize()
        print("n=%4d %-18s %.1f us/step" % (n, "+".join(want), (time.perf_counter() - t0) / 1000 * 1e6))
