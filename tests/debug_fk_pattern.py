"""Development aid (not collected by pytest): where does the GPU FK differ from the CPU oracle?  usage: python tests/debug_fk_pattern.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smplpp_amd import model_io
from smplpp_amd.smpl import SMPL
from oracle.cpu import OracleModel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 70
md = model_io.synthetic_model()
s = SMPL(); s.setDevice("cuda:0"); s.init(md)
o = OracleModel(md)
beta, theta = model_io.synthetic_inputs(n, seed=3)
g = s.launch(beta, theta)
r = o.fk(beta, theta)
for key in ("rest", "verts"):
    err = np.abs(g[key] - r[key]).max(axis=2)  # [n, V]
    bad = err > 1e-5
    print(key, "max err", err.max(), "bad count", int(bad.sum()), "of", bad.size)
    if bad.any():
        fr = np.where(bad.any(axis=1))[0]; vs = np.where(bad.any(axis=0))[0]
        print("  bad frames:", fr[:40], "..." if len(fr) > 40 else "")
        print("  bad verts (first 80):", vs[:80])
        print("  bad vert groups of 32:", sorted(set((vs // 32).tolist()))[:60])
        print("  bad per frame:", bad.sum(axis=1)[:70])
W = md["weights"] if "weights" in md else md["W"]
W = np.asarray(W).reshape(-1, 24)
err = np.abs(g["verts"] - r["verts"]).max(axis=2)
for f in (0, 2, 4, 8, 33, 64):
    if f >= n: continue
    badv = np.where(err[f] > 1e-5)[0]; goodv = np.where(err[f] <= 1e-5)[0]
    goodj = set(np.where((W[goodv] != 0).any(axis=0))[0].tolist())
    badj = set(np.where((W[badv] != 0).any(axis=0))[0].tolist())
    print("frame", f, "culprit joints:", sorted(badj - goodj), " good joints:", sorted(goodj))
gv, rv = g["verts"], r["verts"]
for f, v in ((0, 12), (0, 13), (2, 28), (8, 44)):
    d = np.abs(rv[:, v, :] - gv[f, v, :]).max(axis=1)
    print("gpu verts[%d,%d] =" % (f, v), gv[f, v], "oracle", rv[f, v], " closest oracle frame:", int(d.argmin()), "dist", d.min())
    # is it the un-translated / unskinned value?
    print("     rest", r["rest"][f, v], " root", theta[f, 0])
