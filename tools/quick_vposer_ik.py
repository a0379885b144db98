"""Development aid: the configs[4] leg alone (VPoser-latent IK, 512 frames x 6 position targets x 50 iterations)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplpp_amd import model_io
from smplpp_amd.smpl import SMPL
from smplpp_amd.ik import IkSolver, VPoserDecoder, reference_task_faces
n, K = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 6
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
model = model_io.synthetic_model()
s = SMPL(); s.setDevice("cuda:0"); s.init(model)
vp = VPoserDecoder(VPoserDecoder.synthetic_params())
_, faces = reference_task_faces(K)
rng = np.random.default_rng(300)
hid = np.zeros((n, 25, 3), np.float32); hid[:, 1:22] = rng.normal(0, 0.15, (n, 21, 3))
hv = s.launch(np.zeros((n, 10), np.float32), hid, want=("verts",))["verts"]
tp = hv[:, model["face_indices"][faces] - 1].mean(axis=2)
sol = IkSolver(s, n, K, vposer=vp)
sol.setTasks(face_idx=faces, target_pos=tp, phi_limit=np.zeros(K), normal_task_weight=np.zeros(K))
best = 1e9
for rep in range(4):
    sol.setTasks(face_idx=faces, vertex_weights=np.full((K, 3), 1 / 3, np.float32))
    sol.setConfig(np.zeros((n, 10), np.float32), np.zeros((n, 44), np.float32))
    torch.cuda.synchronize(); t = time.perf_counter()
    e2 = sol.iterate(iters)
    torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
print("VPoser IK %dx%dx%d: %.1f us per iteration-batch, %.2f M it/s" % (n, K, iters, best / iters * 1e6, n * iters / best / 1e6))
import ctypes
from smplpp_amd import _lib
_eq = ctypes.c_double(0.0); _L = _lib.load()
_L.smplpp_debug_ik_enqueue_us.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
_L.smplpp_debug_ik_enqueue_us(sol._h, ctypes.byref(_eq))
print("host enqueue of the last call: %.1f us per iteration" % (_eq.value / iters))
