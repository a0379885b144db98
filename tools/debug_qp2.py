import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smplpp_amd import model_io
from smplpp_amd.smpl import SMPL
from smplpp_amd.ik import IkSolver
g = np.load('tests/golden/ik_synth.npz')
m = model_io.synthetic_model()
s = SMPL(); s.setDevice("cuda:0"); s.init(m)
K = len(g["face_idx"])
kw = dict(face_idx=g["face_idx"], target_pos=g["target_pos"], target_normal=g["target_normal"],
          phi_limit=np.full(K, 0.04), normal_offset=np.full(K, 0.015), normal_task_weight=np.zeros(K))
def fresh():
    sol = IkSolver(s, 1, K); sol.setTasks(**kw); sol.setConfig(g["beta"][None], g["traj_theta"][0][None]); return sol
a = fresh(); a.iterate(4, enable_qp=True, optimize_beta_from=2); ba, ta = a.getConfig()
b = fresh()
for it in range(4): b.iterate(1, enable_qp=True, optimize_beta_from=(0 if it >= 2 else 1000))
bb, tb = b.getConfig()
print("loop4 vs 4x1:", np.abs(ta-tb).max(), np.abs(ba-bb).max())
c = fresh(); c.iterate(2, enable_qp=True, optimize_beta_from=2); c.iterate(2, enable_qp=True, optimize_beta_from=0); bc, tc = c.getConfig()
print("2+2 vs 4x1:", np.abs(tc-tb).max(), np.abs(bc-bb).max())
d = fresh(); d.iterate(3, enable_qp=True, optimize_beta_from=2); bd, td = d.getConfig()
e = fresh()
for it in range(3): e.iterate(1, enable_qp=True, optimize_beta_from=(0 if it >= 2 else 1000))
be, te = e.getConfig()
print("loop3 vs 3x1:", np.abs(td-te).max())
