"""Development aid: configs[3] alone — every frame of sample_walk.c3d, R restarts, direct theta or (third argument `latent`) the
44-d VPoser layout the reference forces on capture solves (for rocprofv3 --kernel-trace --stats).
usage: python tools/mocap_full.py [R] [frames] [latent]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplpp_amd import model_io, mocap
from smplpp_amd.smpl import SMPL
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = SMPL(); s.setDevice("cuda:0"); s.init(model_io.synthetic_model())
g = np.load(os.path.join(ROOT, "tests", "golden", "sample_walk_full.npz"))
names = list(g["task_names"]); faces = np.array([mocap.BASELINE41[nm] for nm in names], np.int64); K = len(names)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else g["points"].shape[0]
pts = (g["points"][:T] - g["points"][0][g["valid"][0]].mean(axis=0) + np.array([0, -0.3, 0], np.float32)).astype(np.float32)
valid = g["valid"][:T]
rng = np.random.default_rng(200)
th0 = np.zeros((R, 25, 3), np.float32); th0[:, 1:] = rng.normal(0, 0.03, (R, 24, 3))
latent = len(sys.argv) > 3 and sys.argv[3] == "latent"
vp = None
if latent:
    from smplpp_amd.ik import VPoserDecoder
    vp = VPoserDecoder(VPoserDecoder.synthetic_params(seed=3))
    th0 = np.zeros((R, 44), np.float32); th0[:, 6:38] = rng.normal(0, 0.05, (R, 32))
ms = mocap.MocapMotionSolver(s, faces, np.full((K, 3), 1 / 3, np.float32), restarts=R, vposer=vp)
ms.solve(pts, valid, np.zeros(10, np.float32), th0, max_frames=2)
torch.cuda.synchronize(); t = time.perf_counter()
th, fr = ms.solve(pts, valid, np.zeros(10, np.float32), th0)
torch.cuda.synchronize(); dt = time.perf_counter() - t
iters = mocap.MocapMotionSolver.WARMUP_ITERS + T - 1
import ctypes
from smplpp_amd import _lib
eq = ctypes.c_double(0.0)
L = _lib.load()
if hasattr(L, "smplpp_debug_ik_enqueue_us"):
    L.smplpp_debug_ik_enqueue_us.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
    L.smplpp_debug_ik_enqueue_us(ms.solver._h, ctypes.byref(eq))
    print("host enqueue: %.1f ms = %.1f us per iteration" % (eq.value / 1e3, eq.value / (mocap.MocapMotionSolver.WARMUP_ITERS + T - 1)))
print(("latent layout, " if latent else "") + "R=%d: %d frames (%d iterations) in %.1f ms -> %.0f solved frames/s, %.1f us per iteration" % (R, T, iters, dt * 1e3, R * T / dt, dt / iters * 1e6))
