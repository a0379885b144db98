"""Development aid: FK throughput with TWO model handles (each its own workspace) on two streams, steps alternating between them —
the double-buffered pipeline a caller streaming batches would run — against the same steps on one handle and one stream.
usage: python tools/fk_two_streams.py [frames] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplpp_amd import model_io
from smplpp_amd.smpl import SMPL

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
H = int(sys.argv[3]) if len(sys.argv) > 3 else 2
model = model_io.synthetic_model()
hs = []
for i in range(H):
    s = SMPL(); s.setDevice("cuda:0"); s.init(model); hs.append(s)
b, t = model_io.synthetic_inputs(n)
bd, td = torch.from_numpy(b).cuda(), torch.from_numpy(t).cuda()
outs = [{"verts": torch.empty((n, 6890, 3), dtype=torch.float32, device="cuda")} for _ in range(H)]
sts = [torch.cuda.Stream() for _ in range(H)]
def run(k, two):
    for i in range(k):
        j = i % H if two else 0
        with torch.cuda.stream(sts[j]):
            hs[j].launch(bd, td, want=("verts",), out=outs[j])
for two in (False, True, False, True):
    run(600, two); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(steps, two); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s: n=%d  %.2f us/step  %.3g evals/s" % (("%d handles, %d streams" % (H, H)) if two else "one handle, one stream  ", n, dt / steps * 1e6, n * steps / dt))
ref = hs[0].launch(bd, td, want=("verts",))["verts"]
torch.cuda.synchronize()
print("outputs equal:", bool((outs[0]["verts"] == ref).all() and (outs[H - 1]["verts"] == ref).all()))
