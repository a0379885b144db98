"""Development aid: phase timestamps of ik_eval_kernel for the configs[2] workload (needs a library built with
-DSMPLPP_EVAL_STAMPS: tools/build_variant.sh stamps ik.hip -DSMPLPP_EVAL_STAMPS ; SMPLPP_HIP_LIB=$PWD/ab/stamps.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np, torch
from smplpp_amd import model_io, _lib
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
nw = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
sol = IkSolver(s, n, K)
sol.setTasks(face_idx=faces, target_pos=tp, target_normal=tn, phi_limit=np.zeros(K), normal_task_weight=np.full(K, nw))
sol.setConfig(np.zeros((n, 10), np.float32), th0)
sol.iterate(int(sys.argv[2]) if len(sys.argv) > 2 else 10)
L = _lib.load(); buf = (ctypes.c_ulonglong * (64 * 16))()
L.smplpp_debug_eval_stamps.restype = ctypes.c_int
assert L.smplpp_debug_eval_stamps(buf) == 0
T = np.array(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64)
names = ["const", "chain", "A0", "A1", "A2", "A3", "B(all)"]
d = np.diff(T[:, :8], axis=1)
print("ticks (100 MHz): " + "  ".join("%s %d" % (nm, np.median(d[:, i])) for i, nm in enumerate(names)), " total", np.median(T[:, 7] - T[:, 0]))
g = T[:, 8:14]
tot = T[:, 7] - T[:, 0]
print("total per workgroup: median %d  p90 %d  max %d ; B(all) median %d max %d ; A0 median %d max %d" % (np.median(tot), np.percentile(tot, 90), tot.max(), np.median(d[:, 6]), d[:, 6].max(), np.median(d[:, 2]), d[:, 2].max()))
print("first group: tables %d  B1 %d  B2 %d  B3n %d  B3 %d  B4 %d" % tuple([np.median(g[:, 0] - T[:, 6])] + [np.median(g[:, i + 1] - g[:, i]) for i in range(5)]))
