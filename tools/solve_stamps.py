"""Development aid: phase timestamps (100 MHz) of the primal ik_solve_kernel in the capture fit (needs a library built with
-DSMPLPP_SOLVE_STAMPS: tools/build_variant.sh sstamps ik.hip -DSMPLPP_SOLVE_STAMPS ; SMPLPP_HIP_LIB=$PWD/ab/sstamps.so).
usage: python tools/solve_stamps.py [R | ik] [latent]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smplpp_amd import model_io, mocap, _lib
from smplpp_amd.smpl import SMPL
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = SMPL(); s.setDevice("cuda:0"); s.init(model_io.synthetic_model())
g = np.load(os.path.join(ROOT, "tests", "golden", "sample_walk_excerpt.npz"))
names = list(g["task_names"]); faces = np.array([mocap.BASELINE41[nm] for nm in names], np.int64); K = len(names)
pts = g["points"] - g["points"][0][g["valid"][0]].mean(axis=0) + np.array([0, -0.3, 0], np.float32)
if len(sys.argv) > 1 and sys.argv[1] == "ik":  # configs[2]: the dual form (24 residual rows against 75 unknowns)
    from smplpp_amd.ik import IkSolver, reference_task_faces
    n, K = 256, 6
    _, fc = reference_task_faces(K)
    rng = np.random.default_rng(100)
    tp = rng.normal(0, 0.3, (n, K, 3)).astype(np.float32)
    th = np.zeros((n, 25, 3), np.float32); th[:, 1:] = rng.normal(0, 0.05, (n, 24, 3))
    sol = IkSolver(s, n, K)
    sol.setTasks(face_idx=fc, target_pos=tp, phi_limit=np.zeros(K))
    sol.setConfig(np.zeros((n, 10), np.float32), th)
    sol.iterate(10)
    L = _lib.load(); buf = (ctypes.c_ulonglong * (64 * 16))()
    L.smplpp_debug_solve_stamps.restype = ctypes.c_int
    assert L.smplpp_debug_solve_stamps(buf) == 0
    T = np.array(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64)
    seq = [0, 1, 2, 3, 4, 5, 11]
    nm = ["set-up", "lists + gather", "column scaling", "Gram r x r", "factorisation + substitutions (one wavefront)", "x + update"]
    d = np.diff(T[:, seq], axis=1) * 0.01
    print("dual solve, us per phase (median over 64 workgroups): " + "  ".join("%s %.2f" % (a, b) for a, b in zip(nm, np.median(d, axis=0))), " total %.1f" % np.median((T[:, 11] - T[:, 0]) * 0.01))
    sys.exit(0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
th0 = np.zeros((R, 25, 3), np.float32)
vp = None
if len(sys.argv) > 2 and sys.argv[2] == "latent":  # the 44-d VPoser layout: 44 free unknowns, the prior on the diagonal
    from smplpp_amd.ik import VPoserDecoder
    vp = VPoserDecoder(VPoserDecoder.synthetic_params(seed=3))
    th0 = np.zeros((R, 44), np.float32); th0[:, 6:38] = np.random.default_rng(200).normal(0, 0.05, (R, 32))
ms = mocap.MocapMotionSolver(s, faces, np.full((K, 3), 1 / 3, np.float32), restarts=R, vposer=vp)
ms.solve(pts, g["valid"], np.zeros(10, np.float32), th0, max_frames=40)
L = _lib.load(); buf = (ctypes.c_ulonglong * (64 * 16))()
L.smplpp_debug_solve_stamps.restype = ctypes.c_int
assert L.smplpp_debug_solve_stamps(buf) == 0
TT = np.array(buf, dtype=np.uint64).reshape(64, 16)[:min(R, 64)].astype(np.int64)
print("between set-up and staging: free-set list %.2f us, rowv %.2f us, into the build (tile columns, W) %.2f us" % tuple(np.median(x) * 0.01 for x in (TT[:, 14] - TT[:, 1], TT[:, 15] - TT[:, 14], TT[:, 2] - TT[:, 15])))
print("staging: addresses %.2f us, issue %.2f us, wait %.2f us" % tuple(np.median(x) * 0.01 for x in (TT[:, 12] - TT[:, 3], TT[:, 13] - TT[:, 12], TT[:, 4] - TT[:, 13])))
T = TT[:, :12]
nm = ["set-up", "lists+rowv+tiles", "barrier", "voff+DMA+wait", "barrier", "Gram", "tiles->regs", "factorisation", "pivots+scale", "back subst", "QP tail+update"]
d = np.diff(T, axis=1) * 0.01
print("us per phase (median over %d workgroups of the last launch): " % len(T) + "  ".join("%s %.2f" % (a, b) for a, b in zip(nm, np.median(d, axis=0))), " total %.1f" % np.median((T[:, 11] - T[:, 0]) * 0.01))
