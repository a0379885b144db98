"""Development aid: error of the fused kernel's forms against float64, on the synthetic model (batch 200, |beta| x 3) and at real-SMPL
magnitudes (posedirs to 5e-2, |beta| = 3, 0.9 rad rotations, 5 m root offsets: tests/test_fk_gpu.py), and their step time.
usage (GPU box): python3 tools/fk_form_errors.py [forms, default "h b"]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.spatial.transform import Rotation
from smplpp_amd import model_io
from smplpp_amd.smpl import SMPL


def f64_reference(md, beta, theta, o):
    n = len(beta)
    th = theta[:, 1:].astype(np.float64) + 1e-8
    ang = np.linalg.norm(th, axis=-1, keepdims=True)
    axis = theta[:, 1:].astype(np.float64) / ang
    R = Rotation.from_rotvec((axis * ang).reshape(-1, 3)).as_matrix().reshape(n, 24, 3, 3)
    c = (R[:, 1:] - np.eye(3)).reshape(n, 207)
    rest = (md["vertices_template"].astype(np.float64)[None] + np.einsum("vxk,nk->nvx", md["shape_blend_shapes"].astype(np.float64), beta.astype(np.float64))
            + np.einsum("vxk,nk->nvx", md["pose_blend_shapes"].astype(np.float64), c))
    G = o["xforms"].astype(np.float64)
    W = md["weights"].astype(np.float64)
    M = np.einsum("vj,njab->nvab", W, G)
    h = np.einsum("nvab,nvb->nva", M[:, :, :3, :3], o["rest"].astype(np.float64)) + M[:, :, :3, 3]
    verts = h / W.sum(axis=1)[None, :, None] + theta[:, :1].astype(np.float64)
    return rest, verts


forms = (sys.argv[1] if len(sys.argv) > 1 else "h b").split()
base = model_io.synthetic_model()
rng = np.random.default_rng(77)
stress = {k: v.copy() for k, v in base.items()}
stress["pose_blend_shapes"] = np.clip(rng.normal(0, 5e-2 / 3, base["pose_blend_shapes"].shape), -5e-2, 5e-2).astype(np.float32)
n = 96
beta_s = rng.choice([-3.0, 3.0], size=(n, 10)).astype(np.float32)
theta_s = np.zeros((n, 25, 3), np.float32)
theta_s[:, 1:] = rng.normal(0, 0.9, (n, 24, 3))
theta_s[:, 0] = rng.uniform(-5, 5, (n, 3))
beta_b, theta_b = model_io.synthetic_inputs(200, seed=11)
beta_b = (beta_b * 3.0).astype(np.float32)
bt, tt = model_io.synthetic_inputs(1024)
btd, ttd = torch.from_numpy(bt).cuda(), torch.from_numpy(tt).cuda()
for form in forms:
    os.environ["SMPLPP_SKIN"] = form
    line = "form %s:" % form
    for name, md, b, t in (("synthetic", base, beta_b, theta_b), ("real-SMPL magnitudes", stress, beta_s, theta_s)):
        s = SMPL(); s.setDevice("cuda:0"); s.init(md)
        o = s.launch(b, t)
        rest, verts = f64_reference(md, b, t, o)
        line += "  %s: rest %.3g  verts %.3g m" % (name, np.abs(o["rest"] - rest).max(), np.abs(o["verts"] - verts).max())
    s = SMPL(); s.setDevice("cuda:0"); s.init(base)
    for _ in range(600): s.launch(btd, ttd, want=("verts",))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): s.launch(btd, ttd, want=("verts",))
    e1.record(); torch.cuda.synchronize()
    line += "   batch 1024: %.2f us per step" % (e0.elapsed_time(e1) / 200 * 1e3)
    print(line)
