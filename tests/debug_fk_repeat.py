"""Development aid: run-to-run determinism of the fused kernel + error vs oracle on all 1024 frames."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smplpp_amd import model_io
from smplpp_amd.smpl import SMPL
from oracle.cpu import OracleModel
md = model_io.synthetic_model()
s = SMPL(); s.setDevice("cuda:0"); s.init(md)
beta, theta = model_io.synthetic_inputs(1024, seed=1)
r = OracleModel(md).fk(beta, theta, want=("verts",))["verts"]
for it in range(4):
    o = s.launch(beta, theta, want=("verts",))["verts"]
    dd = np.abs(o - r); ww = np.argwhere(dd > 1e-5)
    print("run", it, "max err", dd.max(), "n bad", len(ww), "frames%32", sorted(set((ww[:, 0] % 32).tolist()))[:20], "comps", sorted(set(ww[:, 2].tolist())), "v%16", sorted(set((ww[:, 1] % 16).tolist())))
import collections
t = (ww[:, 1] // 64) * 16 + (ww[:, 0] // 64)
print("bad by item position in block (t % 7):", sorted(collections.Counter((t % 7).tolist()).items()))
print("bad by ftp:", sorted(collections.Counter((ww[:, 0] // 64).tolist()).items()))
print("bad by wave-half of frames (f%64//32):", sorted(collections.Counter(((ww[:, 0] % 64) // 32).tolist()).items()), " by v%64//32:", sorted(collections.Counter(((ww[:, 1] % 64) // 32).tolist()).items()))
print("distinct items bad:", len(set(t.tolist())), "of", 108 * 16)
# per bad item: how many of its lanes are bad
per = collections.Counter(t.tolist()); print("bad count per item (sample):", list(per.items())[:12])
