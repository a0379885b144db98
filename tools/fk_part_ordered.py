"""Development aid (round 5): what the skinning-class groups of skin_kernel_h are worth on a model whose VERTEX ORDER follows the body
parts, as SMPL's does (the synthetic stand-in's follows a spiral over an ellipsoid: 11 of its 108 groups of 64 consecutive vertices
are single-class).  The same synthetic model with its vertices RELABELLED — sorted by which k-step of the skinning product their
weights touch (joints 0..15 | both | joints 16..23), every array and the faces permuted consistently — is the same body with another
numbering; timed in both numberings, checked against the oracle in the new one.
usage (GPU box): python3 tools/fk_part_ordered.py      [SMPLPP_HIP_LIB=$PWD/ab/<variant>.so for another build]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplpp_amd import model_io
from smplpp_amd.smpl import SMPL


relabel, classes = model_io.relabel_vertices, model_io.skinning_classes


def step_us(model, n=1024, steps=2000):
    s = SMPL(); s.setDevice("cuda:0"); s.init(model)
    b, t = model_io.synthetic_inputs(n)
    bd, td = torch.from_numpy(b).cuda(), torch.from_numpy(t).cuda()
    for _ in range(600): s.launch(bd, td, want=("verts",))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): s.launch(bd, td, want=("verts",))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3, s


base = model_io.synthetic_model()
cls = classes(base["weights"])
order = np.argsort(cls, kind="stable")
part = relabel(base, order)
for name, m in (("spiral numbering (the stand-in as it is)", base), ("part-ordered numbering", part)):
    c = classes(m["weights"])
    V = len(c); nt = (V + 63) // 64
    g = [(int((c[t * 64:(t + 1) * 64] != 2).any()) | 2 * int((c[t * 64:(t + 1) * 64] != 0).any())) for t in range(nt)]
    us, s = step_us(m)
    print("%-42s groups: joints 0..15 only %3d, both %3d, joints 16..23 only %3d   batch 1024: %.2f us per step" % (name, g.count(1), g.count(3), g.count(2), us))
# parity of the relabelled model against the oracle (and against the original numbering: the same body)
from oracle import cpu
b, t = model_io.synthetic_inputs(40, seed=9)
s = SMPL(); s.setDevice("cuda:0"); s.init(part)
o = s.launch(b, t)
r = cpu.OracleModel(part).fk(b, t)
s0 = SMPL(); s0.setDevice("cuda:0"); s0.init(base)
o0 = s0.launch(b, t)
print("part-ordered model vs oracle: verts %.3g m, rest %.3g m; vs the original numbering (same body, vertices matched): %s" % (
    np.abs(o["verts"] - r["verts"]).max(), np.abs(o["rest"] - r["rest"]).max(),
    "bit-identical" if np.array_equal(o["verts"], o0["verts"][:, order]) else "max diff %.3g" % np.abs(o["verts"] - o0["verts"][:, order]).max()))
