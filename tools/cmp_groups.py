import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from smplpp_amd import model_io
from smplpp_amd.smpl import SMPL
from smplpp_amd.ik import IkSolver, reference_task_faces
n, K = 256, 6
model = model_io.synthetic_model()
s = SMPL(); s.setDevice("cuda:0"); s.init(model)
_, faces = reference_task_faces(K)
rng = np.random.default_rng(100)
hid = np.zeros((n, 25, 3), np.float32); hid[:, 1:] = rng.normal(0, 0.2, (n, 24, 3))
hv = s.launch(np.zeros((n, 10), np.float32), hid, want=("verts",))["verts"]
f0 = model["face_indices"][faces] - 1
tp = hv[:, f0].mean(axis=2)
tn = s.calcVertexNormalBatch(f0.reshape(-1)).reshape(n, K, 3, 3).mean(axis=2)
tn = -(tn / np.linalg.norm(tn, axis=-1, keepdims=True)).astype(np.float32)
th0 = np.zeros((n, 25, 3), np.float32); th0[:, 1:] = rng.normal(0, 0.05, (n, 24, 3))
res = []
for env in ("40", None):
    if env: os.environ["SMPLPP_IK_DBG_STOP"] = env
    else: os.environ.pop("SMPLPP_IK_DBG_STOP", None)
    sol = IkSolver(s, n, K)
    sol.setTasks(face_idx=faces, target_pos=tp, target_normal=tn, phi_limit=np.zeros(K), normal_task_weight=np.ones(K))
    sol.setConfig(np.zeros((n, 10), np.float32), th0)
    e, J = sol.eval()
    e2 = sol.iterate(50)
    res.append((e, J, e2, sol.getConfig()[1]))
print("eval e diff", np.abs(res[0][0] - res[1][0]).max(), "J diff", np.abs(res[0][1] - res[1][1]).max())
print("e2 single", np.sort(res[0][2])[-5:], "pair", np.sort(res[1][2])[-5:])
print("frames with e2 > 1e-3:", int((res[1][2] > 1e-3).sum()), "of", n, "; median", float(np.median(res[1][2])))
print("theta diff", np.abs(res[0][3] - res[1][3]).max())
