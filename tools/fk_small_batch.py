"""Development aid: FK step time at small batches, with and without the rest-shape output (the IK loop asks for both:
its evaluation reads the rest positions of the ring vertices).  usage: python tools/fk_small_batch.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplpp_amd import model_io
from smplpp_amd.smpl import SMPL

s = SMPL(); s.setDevice("cuda:0"); s.init(model_io.synthetic_model())
for n in (64, 128, 256, 512, 1024):
    b, t = model_io.synthetic_inputs(n)
    bd, td = torch.from_numpy(b).cuda(), torch.from_numpy(t).cuda()
    for want in (("verts",), ("verts", "rest")):
        out = {k: torch.empty(n, 6890, 3, device="cuda") for k in want}
        for _ in range(600): s.launch(bd, td, want=want, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(400): s.launch(bd, td, want=want, out=out)
        e1.record(); torch.cuda.synchronize()
        print("n=%4d  want=%-18s %.2f us/step" % (n, "+".join(want), e0.elapsed_time(e1) / 400 * 1e3))
