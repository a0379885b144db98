"""Model files either side of the hot path.

* `load_model(path)` reads the reference's on-disk schema — the seven arrays written by
  /root/reference/scripts/preprocess.py:98-117 (`vertices_template`, `face_indices` (1-based, :91), `weights`,
  `shape_blend_shapes`, `pose_blend_shapes`, `joint_regressor`, `kinematic_tree`) from either the `.npz` or the
  `.json` it produces (the `.json` is what SMPL::init parses, src/SMPL.cpp:560-612).
* `synthetic_model()` builds a deterministic stand-in of the real shapes: the SMPL parameter files are
  license-gated and absent from the reference repository (README.md:22-26), and there is no network.

Everything here is host-side numpy; nothing touches the GPU.
"""
from __future__ import annotations

import json
import os
from typing import Dict

import numpy as np

VERTEX_NUM = 6890  # include/smplpp/definition/def.h:9
JOINT_NUM = 24  # def.h:10
SHAPE_BASIS_DIM = 10  # def.h:11
POSE_BASIS_DIM = 207  # def.h:12
FACE_INDEX_NUM = 13776  # def.h:13
LATENT_DIM = 32  # def.h:14

# The real SMPL kinematic tree (src/toolbox/Tester.cpp:721-723); row 0 = parent, row 1 = joint id.
KINEMATIC_TREE = np.array(
    [
        [4294967295, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21],
        list(range(24)),
    ],
    dtype=np.int64,
)

MODEL_KEYS = (
    "vertices_template",
    "face_indices",
    "weights",
    "shape_blend_shapes",
    "pose_blend_shapes",
    "joint_regressor",
    "kinematic_tree",
)

_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

# Approximate SMPL rest-pose joint centres (metres, Y up), only used to place the synthetic skeleton.
_REST_JOINTS = np.array(
    [
        [0.00, -0.22, 0.03], [0.07, -0.31, 0.02], [-0.07, -0.31, 0.02], [0.00, -0.10, 0.00],
        [0.10, -0.69, 0.02], [-0.10, -0.69, 0.02], [0.00, 0.04, 0.02], [0.09, -1.09, -0.02],
        [-0.09, -1.09, -0.02], [0.00, 0.09, 0.04], [0.12, -1.15, 0.10], [-0.12, -1.15, 0.10],
        [0.00, 0.30, 0.00], [0.08, 0.21, 0.00], [-0.08, 0.21, 0.00], [0.00, 0.38, 0.04],
        [0.18, 0.24, 0.00], [-0.18, 0.24, 0.00], [0.44, 0.23, -0.02], [-0.44, 0.23, -0.02],
        [0.69, 0.24, -0.01], [-0.69, 0.24, -0.01], [0.78, 0.23, -0.01], [-0.78, 0.23, -0.01],
    ],
    dtype=np.float64,
)  # fmt: skip


def _normalise(model: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Cast to the dtypes SMPL::init uses (src/SMPL.cpp:572-612) and check the shapes it checks."""
    out = {
        "vertices_template": np.ascontiguousarray(model["vertices_template"], dtype=np.float32),
        "face_indices": np.ascontiguousarray(model["face_indices"], dtype=np.int32),
        "weights": np.ascontiguousarray(model["weights"], dtype=np.float32),
        "shape_blend_shapes": np.ascontiguousarray(model["shape_blend_shapes"], dtype=np.float32),
        "pose_blend_shapes": np.ascontiguousarray(model["pose_blend_shapes"], dtype=np.float32),
        "joint_regressor": np.ascontiguousarray(model["joint_regressor"], dtype=np.float32),
        "kinematic_tree": np.ascontiguousarray(model["kinematic_tree"], dtype=np.int64),
    }
    v = out["vertices_template"].shape[0]
    if out["shape_blend_shapes"].shape != (v, 3, SHAPE_BASIS_DIM):
        # message follows src/SMPL.cpp:585-586
        raise ValueError(
            "Shape parameter dimensions are invalid: %d != %d" % (out["shape_blend_shapes"].shape[-1], SHAPE_BASIS_DIM)
        )
    if out["pose_blend_shapes"].shape != (v, 3, POSE_BASIS_DIM):
        raise ValueError(
            "Pose parameter dimensions are invalid: %d != %d" % (out["pose_blend_shapes"].shape[-1], POSE_BASIS_DIM)
        )
    if out["vertices_template"].shape != (v, 3) or out["weights"].shape != (v, JOINT_NUM):
        raise ValueError("Cannot initialize a SMPL model!")
    if out["joint_regressor"].shape != (JOINT_NUM, v) or out["kinematic_tree"].shape != (2, JOINT_NUM):
        raise ValueError("Cannot initialize a SMPL model!")
    if out["face_indices"].ndim != 2 or out["face_indices"].shape[1] != 3:
        raise ValueError("Cannot initialize a SMPL model!")
    if out["face_indices"].min() < 1 or out["face_indices"].max() > v:
        raise ValueError("face_indices must be 1-based vertex ids (scripts/preprocess.py:91)")
    return out


def load_model(path: str) -> Dict[str, np.ndarray]:
    """Load a model written by the reference's preprocess.py (`.npz` or `.json`)."""
    if not os.path.exists(path):
        raise FileNotFoundError("Cannot initialize a SMPL model!")  # src/SMPL.cpp:616
    if path.endswith(".npz"):
        with np.load(path) as z:
            model = {k: z[k] for k in MODEL_KEYS}
    else:
        with open(path, "r") as f:
            raw = json.load(f)
        model = {k: np.asarray(raw[k]) for k in MODEL_KEYS}
    return _normalise(model)


def save_model_npz(path: str, model: Dict[str, np.ndarray]) -> None:
    np.savez(path, **{k: model[k] for k in MODEL_KEYS})


def save_model_json(path: str, model: Dict[str, np.ndarray]) -> None:
    with open(path, "w") as f:
        json.dump({k: np.asarray(model[k]).tolist() for k in MODEL_KEYS}, f, indent=4, sort_keys=True)


def _fibonacci_sphere(n: int) -> np.ndarray:
    i = np.arange(n, dtype=np.float64) + 0.5
    phi = np.arccos(1.0 - 2.0 * i / n)
    theta = np.pi * (1.0 + 5.0**0.5) * i
    return np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], axis=1)


_SYNTH_CACHE: Dict[int, Dict[str, np.ndarray]] = {}


def synthetic_model(seed: int = 20250205) -> Dict[str, np.ndarray]:
    """Deterministic SMPL-shaped model (real shapes, real kinematic tree, plausible magnitudes).

    * template: a closed genus-0 surface (a flattened, gently bumpy ellipsoid of human size, Y up) whose
      triangulation (tools/make_synthetic_faces.py) has exactly 6890 vertices / 13776 faces;
    * shapedirs ~ N(0, (0.01/k)^2) for component k = 1..10, posedirs ~ N(0, 0.002^2): dense, like the real ones;
    * joint regressor: 30 non-zeros per joint (15 nearest front + 15 nearest back vertices), Dirichlet weights,
      stored dense like scripts/preprocess.py:95 does;
    * skinning weights: 4 nearest joints per vertex, smooth fall-off, normalised in fp64 THEN cast to fp32 so the
      rows sum to 1 only up to fp32 rounding — the case the reference's homogeneous divide
      (src/LinearBlendSkinning.cpp:545-550) exists for.
    """
    if seed in _SYNTH_CACHE:
        return {k: v.copy() for k, v in _SYNTH_CACHE[seed].items()}
    rng = np.random.default_rng(seed)
    faces0 = np.load(os.path.join(_DATA_DIR, "synthetic_faces.npy")).astype(np.int64)
    assert faces0.shape == (FACE_INDEX_NUM, 3)

    d = _fibonacci_sphere(VERTEX_NUM)
    bump = 1.0 + 0.06 * np.sin(3.0 * d[:, 0] + 1.0) * np.cos(2.0 * d[:, 1]) + 0.04 * np.sin(5.0 * d[:, 2])
    centre = np.array([0.0, -0.36, 0.01])
    half = np.array([0.82, 0.88, 0.14])
    vt = centre + d * half * bump[:, None]

    shapedirs = rng.standard_normal((VERTEX_NUM, 3, SHAPE_BASIS_DIM)) * (0.01 / np.arange(1, SHAPE_BASIS_DIM + 1))
    posedirs = rng.standard_normal((VERTEX_NUM, 3, POSE_BASIS_DIM)) * 0.002

    jreg = np.zeros((JOINT_NUM, VERTEX_NUM), dtype=np.float64)
    front = np.nonzero(vt[:, 2] >= centre[2])[0]
    back = np.nonzero(vt[:, 2] < centre[2])[0]
    for j in range(JOINT_NUM):
        for side in (front, back):
            dist = np.linalg.norm(vt[side] - _REST_JOINTS[j], axis=1)
            sel = side[np.argsort(dist, kind="stable")[:15]]
            jreg[j, sel] = rng.dirichlet(np.ones(15)) * 0.5
    joints = jreg @ vt

    dist = np.linalg.norm(vt[:, None, :] - joints[None, :, :], axis=2)  # [V,24]
    near = np.argsort(dist, axis=1, kind="stable")[:, :4]
    w = np.zeros((VERTEX_NUM, JOINT_NUM), dtype=np.float64)
    dn = np.take_along_axis(dist, near, axis=1)
    wn = np.exp(-((dn / 0.15) ** 2)) + 1e-3
    wn /= wn.sum(axis=1, keepdims=True)
    np.put_along_axis(w, near, wn, axis=1)

    model = _normalise(
        {
            "vertices_template": vt,
            "face_indices": faces0 + 1,
            "weights": w,
            "shape_blend_shapes": shapedirs,
            "pose_blend_shapes": posedirs,
            "joint_regressor": jreg,
            "kinematic_tree": KINEMATIC_TREE.copy(),
        }
    )
    _SYNTH_CACHE[seed] = model
    return {k: v.copy() for k, v in model.items()}


def tiny_model(vertex_num: int, seed: int = 7, faces: np.ndarray | None = None) -> Dict[str, np.ndarray]:
    """A small dense random model (any vertex count) for fast parity cases and ragged-size tests.

    Weights and regressor are fully dense here on purpose (every joint non-zero) so the dense skinning path and
    the sparse one can be checked against each other.
    """
    rng = np.random.default_rng(seed)
    vt = rng.uniform(-0.5, 0.5, (vertex_num, 3))
    w = rng.dirichlet(np.ones(JOINT_NUM), size=vertex_num)
    jreg = rng.dirichlet(np.ones(vertex_num), size=JOINT_NUM)
    if faces is None:
        nf = max(2 * vertex_num - 4, 1)
        faces = np.stack([rng.permutation(vertex_num)[:3] for _ in range(nf)]) + 1
        # make sure every vertex appears in at least one face
        for v in range(vertex_num):
            faces[v % nf, v % 3] = v + 1
        for f in faces:  # no degenerate index triples
            while len(set(f.tolist())) < 3:
                f[rng.integers(3)] = rng.integers(vertex_num) + 1
    return _normalise(
        {
            "vertices_template": vt,
            "face_indices": faces,
            "weights": w,
            "shape_blend_shapes": rng.standard_normal((vertex_num, 3, SHAPE_BASIS_DIM)) * 0.01,
            "pose_blend_shapes": rng.standard_normal((vertex_num, 3, POSE_BASIS_DIM)) * 0.002,
            "joint_regressor": jreg,
            "kinematic_tree": KINEMATIC_TREE.copy(),
        }
    )


def synthetic_inputs(n: int, seed: int = 1):
    """Config-2 style inputs: beta ~ N(0,1); theta rows 1..24 ~ N(0, 0.3^2) rad with a 5 % tail
    of frames at N(0, 1) and one all-zero frame (small-angle path); theta row 0 (root translation) ~ U(-1, 1) m."""
    rng = np.random.default_rng(seed)
    beta = rng.standard_normal((n, SHAPE_BASIS_DIM)).astype(np.float32)
    theta = np.empty((n, JOINT_NUM + 1, 3), dtype=np.float32)
    theta[:, 1:, :] = rng.standard_normal((n, JOINT_NUM, 3)) * 0.3
    tail = rng.random(n) < 0.05
    theta[tail, 1:, :] = rng.standard_normal((int(tail.sum()), JOINT_NUM, 3))
    theta[:, 0, :] = rng.uniform(-1.0, 1.0, (n, 3))
    if n > 1:
        theta[n // 2, 1:, :] = 0.0
    return beta, theta


def relabel_vertices(model: dict, order) -> dict:
    """The same model with vertex order[i] as its vertex i (every per-vertex array, the regressor's columns and the 1-based face
    indices permuted consistently): another numbering of the same body.  Used to build vertex orders that follow the body parts,
    as SMPL's does (tools/fk_part_ordered.py, tests/test_fk_gpu.py)."""
    import numpy as np

    order = np.asarray(order, np.int64)
    inv = np.empty_like(order)
    inv[order] = np.arange(len(order))
    m = {k: v.copy() for k, v in model.items()}
    for k in ("vertices_template", "weights", "shape_blend_shapes", "pose_blend_shapes"):
        m[k] = np.ascontiguousarray(model[k][order])
    m["joint_regressor"] = np.ascontiguousarray(model["joint_regressor"][:, order])
    m["face_indices"] = (inv[model["face_indices"].astype(np.int64) - 1] + 1).astype(model["face_indices"].dtype)
    return m


def skinning_classes(weights):
    """Per vertex: 0 = weights on joints 0..15 only, 1 = on both halves, 2 = on joints 16..23 only (the two k-steps of the fused
    kernel's skinning product: smplpp_amd/csrc/common.h, HB_PERM_OFF)."""
    import numpy as np

    w = np.asarray(weights)
    lo, hi = (w[:, :16] != 0).any(axis=1), (w[:, 16:] != 0).any(axis=1)
    return np.where(hi & lo, 1, np.where(hi, 2, 0))
