"""Stress (not collected by pytest; run by hand on the GPU box): the bf16x3 fused kernel against the fp32-MFMA form over random
batch sizes, and bitwise run-to-run determinism.  usage: python tests/stress_fk_forms.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplpp_amd import model_io
from smplpp_amd.smpl import SMPL

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 120
s = SMPL(); s.setDevice("cuda:0"); s.init(model_io.synthetic_model())
rng = np.random.default_rng(12345)
worst = 0.0
for it in range(rounds):
    n = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 1024, 1025, int(rng.integers(1, 3000))]))
    want = ("verts", "rest") if it % 3 == 0 else ("verts",)
    beta, theta = model_io.synthetic_inputs(n, seed=1000 + it)
    bt, tt = torch.from_numpy(beta).cuda(), torch.from_numpy(theta).cuda()
    os.environ["SMPLPP_SKIN"] = "b"
    a = {k: v.clone() for k, v in s.launch(bt, tt, want=want).items() if k in want}
    b2 = {k: v.clone() for k, v in s.launch(bt, tt, want=want).items() if k in want}
    os.environ["SMPLPP_SKIN"] = "p"
    p = {k: v.clone() for k, v in s.launch(bt, tt, want=want).items() if k in want}
    torch.cuda.synchronize()
    for k in want:
        assert torch.equal(a[k], b2[k]), ("non-deterministic", it, n, k)
        d = float((a[k] - p[k]).abs().max())
        worst = max(worst, d)
        assert d < 2e-6, ("forms differ", it, n, k, d)
    if it % 20 == 0:
        print("round", it, "n", n, "worst |b - p| so far", worst, flush=True)
os.environ.pop("SMPLPP_SKIN", None)
print("STRESS OK: %d rounds, worst |bf16x3 - fp32 MFMA| = %.3g m" % (rounds, worst))
