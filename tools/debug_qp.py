import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smplpp_amd import model_io
from smplpp_amd.smpl import SMPL
from smplpp_amd.ik import IkSolver
from oracle import cpu
g = np.load('tests/golden/ik_synth.npz')
m = model_io.synthetic_model()
s = SMPL(); s.setDevice("cuda:0"); s.init(m)
o = cpu.OracleModel(m)
K = len(g["face_idx"])
kw = dict(face_idx=g["face_idx"], target_pos=g["target_pos"], target_normal=g["target_normal"],
          phi_limit=np.full(K, 0.04), normal_offset=np.full(K, 0.015), normal_task_weight=np.zeros(K))
beta = g["beta"].copy(); theta = g["traj_theta"][0].copy()
ts = cpu.TaskSet(g["face_idx"], g["target_pos"], g["target_normal"], phi_limit=np.full(K, 0.04),
                 normal_offset=np.full(K, 0.015), normal_task_weight=np.zeros(K))
sol = IkSolver(s, 1, K); sol.setTasks(**kw); sol.setConfig(beta[None], theta[None])
for it in range(4):
    frm = 2
    # oracle one iteration from current oracle state, emulating body schedule: use optimize_beta_from = 0 if it>=frm else large
    ob = 0 if it >= frm else 1000
    # GPU state -> compare eval first
    st = sol.getTasks(); gb, gt = sol.getConfig()
    ts_g = cpu.TaskSet(st["face_idx"][0], g["target_pos"], g["target_normal"], phi_limit=np.full(K, 0.04 if it >= frm else 0.0),
                 normal_offset=np.full(K, 0.015), normal_task_weight=np.zeros(K), vertex_weights=st["vertex_weights"][0])
    r = o.ik_eval(gb[0], gt[0], ts_g.copy(), it >= frm)
    e, J = sol.eval(optimize_beta=(it >= frm))
    # NOTE: eval refreshes GPU weights; restore
    sol.setTasks(vertex_weights=st["vertex_weights"])
    Jg = J[0]
    if it < frm:
        Jg = Jg.copy(); Jg[:, 75:75+2*K] = 0  # phi columns dead
    print("it", it, "eval diff e", np.abs(r["e"]-e[0]).max(), "J", np.abs(r["J"]-(Jg if it>=frm else J[0]*0+Jg)).max())
    D = r["J"].shape[1]
    A, b = cpu.normal_equations(r["e"], r["J"], 75, 2*K, 10 if it >= frm else 0)
    lo = np.full(D, -1e30); hi = np.full(D, 1e30)
    pl = 0.04 if it >= frm else 0.0
    lo[75:75+2*K] = -pl; hi[75:75+2*K] = pl
    if it >= frm: lo[75+2*K:] = -0.5; hi[75+2*K:] = 0.5
    x = cpu.box_qp(A, b, lo, hi)
    sol.iterate(1, enable_qp=True, optimize_beta_from=(0 if it >= frm else 1000))
    nb, nt = sol.getConfig()
    dx_gpu = (nt[0] - gt[0]).reshape(-1)
    print("   dtheta diff", np.abs(dx_gpu - x[:75]).max(), "|x|", np.abs(x[:75]).max(), "phi x", x[75:75+2*K].round(4), "beta x", x[75+2*K:].round(3))
    if it >= frm: print("   dbeta gpu", (nb[0]-gb[0]).round(3))
