"""Development aid: the capture-excerpt leg alone (for rocprofv3 --kernel-trace --stats).  usage: python tools/mocap_only.py [R] [latent]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplpp_amd import model_io, mocap
from smplpp_amd.smpl import SMPL
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = SMPL(); s.setDevice("cuda:0"); s.init(model_io.synthetic_model())
g = np.load(os.path.join(ROOT, "tests", "golden", "sample_walk_excerpt.npz"))
names = list(g["task_names"]); faces = np.array([mocap.BASELINE41[nm] for nm in names], np.int64); K = len(names)
pts = g["points"] - g["points"][0][g["valid"][0]].mean(axis=0) + np.array([0, -0.3, 0], np.float32)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(200)
th0 = np.zeros((R, 25, 3), np.float32); th0[:, 1:] = rng.normal(0, 0.03, (R, 24, 3))
vp = None
if len(sys.argv) > 2 and sys.argv[2] == "latent":  # the 44-d VPoser layout (synthetic decoder)
    from smplpp_amd.ik import VPoserDecoder
    vp = VPoserDecoder(VPoserDecoder.synthetic_params(seed=3))
    th0 = np.zeros((R, 44), np.float32); th0[:, 6:38] = rng.normal(0, 0.05, (R, 32))
ms = mocap.MocapMotionSolver(s, faces, np.full((K, 3), 1 / 3, np.float32), restarts=R, vposer=vp)
ms.solve(pts, g["valid"], np.zeros(10, np.float32), th0, max_frames=2)
torch.cuda.synchronize(); t = time.perf_counter()
th, fr = ms.solve(pts, g["valid"], np.zeros(10, np.float32), th0)
torch.cuda.synchronize(); dt = time.perf_counter() - t
print("R=%d: %d frames in %.2f ms -> %.0f solved frames/s" % (R, len(fr), dt * 1e3, R * len(fr) / dt))
if os.environ.get("SMPLPP_HIP_LIB", "").endswith("stamps.so"):
    import ctypes
    from smplpp_amd import _lib
    L = _lib.load(); buf = (ctypes.c_ulonglong * (64 * 16))()
    L.smplpp_debug_eval_stamps.restype = ctypes.c_int
    assert L.smplpp_debug_eval_stamps(buf) == 0
    T = np.array(buf, dtype=np.uint64).reshape(64, 16)[:, :8].astype(np.int64)
    d = np.diff(T, axis=1)
    print("eval phases (ticks, median over 64 workgroups of the LAST launch): const %d chain %d A0 %d A1 %d A2 %d A3 %d B %d ; total %d" % tuple(list(np.median(d, axis=0)) + [np.median(T[:, 7] - T[:, 0])]))
    print("max over workgroups:", d.max(axis=0), (T[:, 7] - T[:, 0]).max())
    TT = np.array(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64)
    g = TT[:, 8:14]
    print("first group of phase B: tables %d  B1 %d  B2 (dp) %d  B3n (normal derivatives) %d  B3 %d  B4 (rows) %d" % tuple([np.median(g[:, 0] - TT[:, 6])] + [np.median(g[:, i + 1] - g[:, i]) for i in range(5)]))
