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
sol = IkSolver(s, 1, K); sol.setTasks(**kw); sol.setConfig(g["beta"][None], g["traj_theta"][0][None])
for it in range(4):
    live = it >= 2
    st = sol.getTasks(); gb, gt = sol.getConfig()
    ts = cpu.TaskSet(st["face_idx"][0], g["target_pos"], g["target_normal"], phi_limit=np.full(K, 0.04),
                 normal_offset=np.full(K, 0.015), normal_task_weight=np.zeros(K), vertex_weights=st["vertex_weights"][0])
    ob, ot, _ = o.ik_solve(gb[0], gt[0], ts, 1, enable_qp=True, optimize_beta_from=(0 if live else 1000))
    sol.iterate(1, enable_qp=True, optimize_beta_from=(0 if live else 1000))
    nb, nt = sol.getConfig(); st2 = sol.getTasks()
    print(it, "theta", np.abs(nt[0]-ot).max(), "beta", np.abs(nb[0]-ob).max(), "faces", st2["face_idx"][0], ts.face_idx,
          "w", np.abs(st2["vertex_weights"][0]-ts.vertex_weights).max())
