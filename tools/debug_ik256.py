import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplpp_amd import model_io
from smplpp_amd.smpl import SMPL
from smplpp_amd.ik import IkSolver, reference_task_faces
def log(*a): print(*a, flush=True)
m = model_io.synthetic_model()
s = SMPL(); s.setDevice("cuda:0"); s.init(m)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = 6
_, faces = reference_task_faces(K)
rng = np.random.default_rng(100)
hid = np.zeros((n, 25, 3), np.float32); hid[:, 1:] = rng.normal(0, 0.2, (n, 24, 3))
hv = s.launch(np.zeros((n, 10), np.float32), hid, want=("verts",))["verts"]
log("fk host ok", np.isfinite(hv).all())
f0 = m["face_indices"][faces] - 1
tp = hv[:, f0].mean(axis=2)
theta0 = np.zeros((n, 25, 3), np.float32); theta0[:, 1:] = rng.normal(0, 0.05, (n, 24, 3))
sol = IkSolver(s, n, K)
sol.setTasks(face_idx=faces, target_pos=tp, phi_limit=np.zeros(K), normal_task_weight=np.zeros(K))
sol.setConfig(np.zeros((n, 10), np.float32), theta0)
log("setup ok")
e, J = sol.eval()
log("eval ok", np.isfinite(e).all(), np.isfinite(J).all(), np.abs(e).max(), np.abs(J).max())
for it in range(3):
    e2 = sol.iterate(1)
    b, t = sol.getConfig()
    st = sol.getTasks()
    log("iter", it, "e2 max", e2.max(), "finite", np.isfinite(t).all(), "faces range", st["face_idx"].min(), st["face_idx"].max())
e2 = sol.iterate(47)
log("iter 50 e2 max", e2.max(), "median", np.median(e2))
