import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
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
sol = IkSolver(s, n, K)
best = 1e9
for rep in range(5):
    sol.setTasks(face_idx=faces, vertex_weights=np.full((K, 3), 1 / 3, np.float32), target_pos=tp, target_normal=tn, phi_limit=np.zeros(K), normal_task_weight=np.ones(K))
    sol.setConfig(np.zeros((n, 10), np.float32), th0)
    torch.cuda.synchronize(); t = time.perf_counter()
    e2 = sol.iterate(50)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    best = min(best, dt)
print("IK 256x6x50: %.1f us per iteration-batch, %.2f M it/s, converged %d" % (best / 50 * 1e6, n * 50 / best / 1e6, int((e2 < 1e-3).sum())))
import ctypes
from smplpp_amd import _lib
_eq = ctypes.c_double(0.0); _L = _lib.load()
_L.smplpp_debug_ik_enqueue_us.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
_L.smplpp_debug_ik_enqueue_us(sol._h, ctypes.byref(_eq))
print("host enqueue of the last call: %.1f us per iteration" % (_eq.value / 50))
