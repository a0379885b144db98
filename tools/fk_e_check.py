"""Development aid: the exact form (skin_e.hip) against round 1's bf16x3 kernel (skin_b.hip) — same piece products in the same
order, same fp32 skinning: the outputs must agree bit for bit — at ragged batch sizes, with and without `rest`, then step times.
usage (GPU box): [FK_CHECK_FORM=x] python3 tools/fk_e_check.py [sizes...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplpp_amd import model_io
from smplpp_amd.smpl import SMPL

sizes = [int(a) for a in sys.argv[1:]] or [1, 63, 64, 65, 200, 1024, 1100]
md = model_io.synthetic_model()
eng = {}
FE = os.environ.get("FK_CHECK_FORM", "e")  # (another form whose bits must equal b's: development builds)
for form in (FE, "b"):
    os.environ["SMPLPP_SKIN"] = form
    s = SMPL(); s.setDevice("cuda:0"); s.init(md)
    eng[form] = s
bad = 0
for n in sizes:
    b, t = model_io.synthetic_inputs(n, seed=5 + n)
    oe = eng[FE].launch(b, t)
    ob = eng["b"].launch(b, t)
    oe2 = eng[FE].launch(b, t, want=("verts",))
    dv, dr = np.abs(oe["verts"] - ob["verts"]).max(), np.abs(oe["rest"] - ob["rest"]).max()
    d2 = np.abs(oe2["verts"] - ob["verts"]).max()
    fin = np.isfinite(oe["verts"]).all()
    print("n=%5d  |e-b| verts %.3g rest %.3g  verts-only launch %.3g  finite %s" % (n, dv, dr, d2, fin), flush=True)
    bad += (dv != 0) or (dr != 0) or (d2 != 0) or not fin
bt, tt = model_io.synthetic_inputs(1024)
btd, ttd = torch.from_numpy(bt).cuda(), torch.from_numpy(tt).cuda()
for form in (FE, "b", FE):
    s = eng[form]
    out = {"verts": torch.empty((1024, 6890, 3), dtype=torch.float32, device="cuda")}
    for _ in range(600): s.launch(btd, ttd, want=("verts",), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): s.launch(btd, ttd, want=("verts",), out=out)
    e1.record(); torch.cuda.synchronize()
    s.profileEnable(True); s.profileRead()
    for _ in range(100): s.launch(btd, ttd, want=("verts",), out=out)
    torch.cuda.synchronize()
    L, kms = s.profileRead(); s.profileEnable(False)
    print("form %s: batch 1024: %.2f us per step, fused kernel %.2f us (%d launches)" % (form, e0.elapsed_time(e1) / 300 * 1e3, kms * 1e3, L), flush=True)
print("MISMATCHES" if bad else "all sizes bit-identical")
sys.exit(1 if bad else 0)
