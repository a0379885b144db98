"""Generate the capture fixtures of BASELINE.json configs[3] from the reference's own data file
(/root/reference/data/sample_walk.c3d, read as DATA with smplpp_amd.mocap.read_c3d — the C3D reader that replaces ezc3d,
node/node.cpp:580-594, 667-690):

  tests/golden/sample_walk_excerpt.npz  32 frames x the 41 Baseline markers (frames chosen to contain missing markers and
                                        frames below the 20-valid-marker skip rule of node/node.cpp:785)
  tests/golden/sample_walk_full.npz     every frame: points [3163, 41, 3] float32 (file units: metres) + valid [3163, 41]

run in the build container (the reference tree does not exist on the GPU box):  python tools/make_sample_walk_fixture.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smplpp_amd import mocap  # noqa: E402

SRC = "/root/reference/data/sample_walk.c3d"
c = mocap.read_c3d(SRC)
names = sorted(mocap.BASELINE41)  # task order = std::map<std::string, IkTask> order (node/node.cpp:47, 798)
idx = mocap.match_markers(c["labels"], names)  # suffix match, node.cpp:587-593
pts = np.ascontiguousarray(c["points"][:, idx])
valid = np.ascontiguousarray(c["valid"][:, idx])
T = pts.shape[0]
lab = np.array([c["labels"][i] for i in idx])
stats = dict(rate=np.float64(c["rate"]), n_frames=np.int64(T), n_points=np.int64(len(c["labels"])),
             missing_labelled=np.int64((~valid).sum()), frames_any_missing=np.int64((~valid).any(axis=1).sum()),
             source=np.array("data/sample_walk.c3d (mmurooka/SMPLpp), units %s" % c["units"]))
out = os.path.join(ROOT, "tests", "golden")
np.savez_compressed(os.path.join(out, "sample_walk_full.npz"), labels=lab, task_names=np.array(names), points=np.where(valid[..., None], pts, 0).astype(np.float32),
                    valid=valid, **stats)
old = os.path.join(out, "sample_walk_excerpt.npz")
if os.path.exists(old):  # keep the committed excerpt's frame choice
    frame_ids = np.load(old)["frame_ids"]
else:
    low = np.where(valid.sum(axis=1) < 20)[0]
    frame_ids = np.concatenate([np.arange(16), low[:16]]) if len(low) >= 16 else np.arange(32)
np.savez_compressed(old, frame_ids=frame_ids, labels=lab, task_names=np.array(names), points=pts[frame_ids], valid=valid[frame_ids], **stats)
nv = valid.sum(axis=1)
print("frames %d, markers %d; frames with a missing marker %d, below 20 valid %d, with 0 valid %d; first low frame %s" % (
    T, len(names), (nv < 41).sum(), (nv < 20).sum(), (nv == 0).sum(), np.where(nv < 20)[0][:5]))
