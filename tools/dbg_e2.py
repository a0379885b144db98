import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smplpp_amd import model_io
from smplpp_amd.smpl import SMPL
from smplpp_amd.ik import IkSolver, reference_task_faces
model = model_io.synthetic_model()
s = SMPL(); s.setDevice("cuda:0"); s.init(model)
for n, K in ((12, 6), (3, 41)):
    _, faces = reference_task_faces(6)
    rng = np.random.default_rng(0)
    faces = rng.integers(0, 13776, K)
    tp = rng.normal(0, 0.3, (n, K, 3)).astype(np.float32)
    th = np.zeros((n, 25, 3), np.float32); th[:, 1:] = rng.normal(0, 0.1, (n, 24, 3))
    sol = IkSolver(s, n, K)
    sol.setTasks(face_idx=faces, target_pos=tp, phi_limit=np.zeros(K))
    sol.setConfig(np.zeros((n, 10), np.float32), th)
    e, J = sol.eval()
    e2 = sol.iterate(1)
    print(n, K, "e2 from solve", e2[:4], "sum e^2", (e ** 2).sum(axis=1)[:4])
    if K == 6:
        q = e[0] ** 2
        print("cumsum", np.cumsum(q))
        print("by task", q.reshape(K, 4).sum(axis=1), "pos only", q.reshape(K, 4)[:, :3].sum())
