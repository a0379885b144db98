"""Development aid (round 5): the evaluation against the oracle over the whole configs[2] run — actual positions, position rows and
normal rows of e, frame by frame after 50 iterations; for the worst frames the oracle's own spread under 1e-7 rad moves of theta
(conditioning yardstick: a deviation inside that spread is the state's conditioning, not the engine's arithmetic).
usage (GPU box): python3 tools/ik_eval_conditioning.py   [SMPLPP_HIP_LIB=$PWD/ab/<variant>.so to look at another build]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import cpu
from smplpp_amd import model_io
from smplpp_amd.smpl import SMPL
from smplpp_amd.ik import IkSolver, reference_task_faces
n, K, iters = 256, 6, 50
model = model_io.synthetic_model()
s = SMPL(); s.setDevice("cuda:0"); s.init(model)
o = cpu.OracleModel(model)
_, faces = reference_task_faces(K)
rng = np.random.default_rng(100)
hid = np.zeros((n, 25, 3), np.float32); hid[:, 1:] = rng.normal(0, 0.2, (n, 24, 3))
hv = s.launch(np.zeros((n, 10), np.float32), hid, want=("verts",))["verts"]
f0 = model["face_indices"][faces] - 1
tp = hv[:, f0].mean(axis=2)
tn = s.calcVertexNormalBatch(f0.reshape(-1)).reshape(n, K, 3, 3).mean(axis=2)
tn = -(tn / np.linalg.norm(tn, axis=-1, keepdims=True)).astype(np.float32)
th0 = np.zeros((n, 25, 3), np.float32); th0[:, 1:] = rng.normal(0, 0.05, (n, 24, 3))
sol = IkSolver(s, n, K)
sol.setTasks(face_idx=faces, target_pos=tp, target_normal=tn, phi_limit=np.zeros(K), normal_task_weight=np.ones(K))
sol.setConfig(np.zeros((n, 10), np.float32), th0)
e2 = sol.iterate(iters)
_, th = sol.getConfig()
tk = sol.getTasks()
e, J = sol.eval()
tk2 = sol.getTasks()
dpos, dn, dap = np.zeros(n), np.zeros(n), np.zeros(n)
for f in range(n):
    ts = cpu.TaskSet(tk["face_idx"][f], tp[f], tn[f], phi_limit=np.zeros(K), vertex_weights=tk["vertex_weights"][f])
    r = o.ik_eval(np.zeros(10, np.float32), th[f].reshape(25, 3), ts)
    de = np.abs(r["e"] - e[f]).reshape(K, 4)
    dpos[f], dn[f] = de[:, :3].max(), de[:, 3].max()
    dap[f] = np.abs(r["actual_pos"] - tk2["actual_pos"][f]).max()
print("max |d actual_pos| %.3g m (median %.3g); position rows max %.3g; normal rows max %.3g median %.3g" % (dap.max(), np.median(dap), dpos.max(), dn.max(), np.median(dn)))
worst = np.argsort(-dn)[:4]
for f in worst:
    ts = cpu.TaskSet(tk["face_idx"][f], tp[f], tn[f], phi_limit=np.zeros(K), vertex_weights=tk["vertex_weights"][f])
    base = o.ik_eval(np.zeros(10, np.float32), th[f].reshape(25, 3), ts)["e"]
    spread = 0.0
    for seed in range(4):
        pert = th[f].reshape(25, 3) + np.random.default_rng(seed).normal(0, 1e-7, (25, 3)).astype(np.float32)
        spread = max(spread, np.abs(o.ik_eval(np.zeros(10, np.float32), pert, ts)["e"] - base).reshape(K, 4)[:, 3].max())
    print("frame %d: |e|^2 %.3g, normal-row deviation engine vs oracle %.3g, oracle's own spread under 1e-7 rad moves %.3g, |d apos| %.3g" % (f, e2[f], dn[f], spread, dap[f]))
