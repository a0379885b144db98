"""Host-side mirror of `smplpp::IkTask` (include/smplpp/IkTask.h:20-84), `smplpp::VPoserDecoder`
(include/smplpp/VPoser.h:53-90) and of the IK loop that lives in the reference's node `main()`
(node/node.cpp:645-1002), batched over independent frames, over the C ABI.

`IkTask` keeps the reference's public field names; `IkSolver` gathers a `dict name -> IkTask` per frame in
`std::map` (lexicographic) order — that order fixes the rows of e/J and the phi column blocks (node.cpp:47,798).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import DEVICE, HOST, SmplppError, check
from .smpl import SMPL, _np32, _ptr

LATENT_DIM = 32
LATENT_POSE_DIM = LATENT_DIM + 12  # node/node.cpp:42


class IkTask:
    """One surface-point task.  Same public fields and defaults as the reference class."""

    def __init__(self, smpl: SMPL, faceIdx: int, targetPos=None, targetNormal=None):
        self.smpl_ = smpl
        self.faceIdx_ = int(faceIdx)
        self.posTaskWeight_ = 1.0
        self.normalTaskWeight_ = 1.0
        self.phiLimit_ = 0.04
        self.normalOffset_ = 0.0
        self.targetPos_ = np.zeros(3, np.float32) if targetPos is None else _np32(targetPos, (3,)).copy()
        self.targetNormal_ = np.array([0, 0, 1], np.float32) if targetNormal is None else _np32(targetNormal, (3,)).copy()
        self.vertexWeights_ = np.full(3, 1.0 / 3.0, np.float32)
        self.tangents_ = np.zeros((3, 2), np.float32)
        self.phi_ = np.zeros(2, np.float32)

    # The four methods evaluate on batch 0 of the SMPL instance's last launch, like the reference (src/IkTask.cpp:33-86);
    # they are thin host compositions of engine queries and exist for API parity — the batched solver does the same
    # arithmetic on the GPU for every frame.
    def _face_vertices(self):
        idx = self.smpl_.getFaceIndexRaw(self.faceIdx_).astype(np.int64) - 1
        v = self.smpl_.getVertexRaw(idx)
        return idx, (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v))

    def calcTangents(self):
        _, fv = self._face_vertices()
        t1 = fv[1] - fv[0]
        n = np.cross(t1, fv[2] - fv[0])
        t2 = np.cross(n, t1)
        t1 = t1 / max(np.linalg.norm(t1), 1e-12)
        t2 = t2 / max(np.linalg.norm(t2), 1e-12)
        self.tangents_ = np.stack([t1, t2], axis=1).astype(np.float32)

    def calcVertexWeights(self, actualPos):
        _, fv = self._face_vertices()
        pos = _np32(actualPos, (3,)) + self.tangents_ @ self.phi_
        w = np.array([np.linalg.norm(np.cross(fv[(i + 1) % 3] - pos, fv[(i + 2) % 3] - pos)) for i in range(3)], np.float32)
        self.vertexWeights_ = (w / w.sum()).astype(np.float32)

    def calcActualNormal(self):
        idx, _ = self._face_vertices()
        vn = self.smpl_.calcVertexNormal(idx)
        vn = vn.detach().cpu().numpy() if hasattr(vn, "detach") else np.asarray(vn)
        n = (self.vertexWeights_[:, None] * vn).sum(axis=0)
        return (n / max(np.linalg.norm(n), 1e-12)).astype(np.float32)

    def calcActualPos(self):
        _, fv = self._face_vertices()
        p = fv.T @ self.vertexWeights_
        if self.normalOffset_ > 0.0:
            p = p + np.float32(self.normalOffset_) * self.calcActualNormal()
        return p.astype(np.float32)


class VPoserDecoder:
    """smplpp::VPoserDecoder: weights in torch::nn::Linear layout, keys as scripts/preprocess_vposer.py:42-47 writes."""

    KEYS = ("decoder_net.0.weight", "decoder_net.0.bias", "decoder_net.3.weight", "decoder_net.3.bias",
            "decoder_net.5.weight", "decoder_net.5.bias")

    def __init__(self, params: Dict[str, np.ndarray], device=0):
        self.params = {k: _np32(params[k]) for k in self.KEYS}
        shapes = [(512, 32), (512,), (512, 512), (512,), (126, 512), (126,)]
        for k, s in zip(self.KEYS, shapes):
            if self.params[k].shape != s:
                raise SmplppError(1, "VPoser parameter %s has shape %s, expected %s" % (k, self.params[k].shape, s))
        _lib.require_gpu()
        h = C.c_void_p()
        check(_lib.load().smplpp_vposer_create(device, *[_ptr(self.params[k]) for k in self.KEYS], C.byref(h)))
        self._h = h

    @classmethod
    def loadParamsFromJson(cls, path, device=0):
        """VPoserDecoderImpl::loadParamsFromJson (src/VPoser.cpp:169-238); `.npz` from the same script also accepted."""
        if path.endswith(".npz"):
            with np.load(path) as z:
                return cls({k: z[k] for k in cls.KEYS}, device)
        import json

        with open(path) as f:
            raw = json.load(f)
        return cls({k: np.asarray(raw[k]) for k in cls.KEYS}, device)

    @staticmethod
    def synthetic_params(seed=3):
        """Stand-in weights (the real VPoser parameters are license-gated and absent): torch::nn::Linear
        default init (U(-1/sqrt(in), 1/sqrt(in))) with a fixed seed, except that the last layer is scaled by 0.3 and
        biased to the identity rotation in the 6D representation ([1,0, 0,1, 0,0] per joint) so that latents near zero
        decode to moderate poses, like a trained pose prior does, instead of uniformly random rotations."""
        rng = np.random.default_rng(seed)
        out = {}
        for (w, b), (o, i) in zip((("decoder_net.0.weight", "decoder_net.0.bias"), ("decoder_net.3.weight", "decoder_net.3.bias"),
                                   ("decoder_net.5.weight", "decoder_net.5.bias")), ((512, 32), (512, 512), (126, 512))):
            k = 1.0 / np.sqrt(i)
            out[w] = rng.uniform(-k, k, (o, i)).astype(np.float32)
            out[b] = rng.uniform(-k, k, (o,)).astype(np.float32)
        out["decoder_net.5.weight"] *= np.float32(0.3)
        out["decoder_net.5.bias"] = (np.float32(0.3) * out["decoder_net.5.bias"]
                                     + np.tile(np.array([1, 0, 0, 1, 0, 0], np.float32), 21)).astype(np.float32)
        return out

    def forward(self, latent, want_jac=False, frame_base=0):
        """`frame_base`: global index of latent[0] when the call decodes a shard of a larger job (same bits as the unsharded call)."""
        z = _np32(latent).reshape(-1, 32)
        n = z.shape[0]
        out = np.empty((n, 21, 3), np.float32)
        jac = np.empty((n, 63, 32), np.float32) if want_jac else None
        check(_lib.load().smplpp_vposer_forward_at(self._h, n, int(frame_base), _ptr(z), _ptr(out), _ptr(jac), HOST, None))
        return (out, jac) if want_jac else out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().smplpp_vposer_destroy(self._h)
                self._h = None
        except Exception:
            pass


def convertRotMatToAxisAngle(rotMat, device=0):
    """smplpp::convertRotMatToAxisAngle (src/VPoser.cpp:25-120)."""
    r = _np32(rotMat).reshape(-1, 3, 3)
    aa = np.empty((r.shape[0], 3), np.float32)
    check(_lib.load().smplpp_rotmat_to_axis_angle(device, r.shape[0], _ptr(r), _ptr(aa), HOST, None))
    return aa


class IkSolver:
    """The loop body of node/node.cpp:645-1002 for `n` independent frames with `K` tasks each."""

    def __init__(self, smpl: SMPL, n: int, K: int, vposer: Optional[VPoserDecoder] = None, frame_base: int = 0):
        """`frame_base`: global index of this solver's frame 0 when it holds one shard of a larger job (dist.shard_range)."""
        self.smpl, self.n, self.K, self.vposer = smpl, int(n), int(K), vposer
        self.theta_dim = LATENT_POSE_DIM if vposer is not None else 75
        h = C.c_void_p()
        check(_lib.load().smplpp_ik_create(smpl.handle, self.n, self.K, vposer._h if vposer is not None else None, C.byref(h)))
        self._h = h
        if frame_base:
            check(_lib.load().smplpp_ik_set_frame_base(self._h, int(frame_base)))
        self.task_names: Optional[List[str]] = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().smplpp_ik_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- tasks
    def setTasks(self, face_idx=None, vertex_weights=None, target_pos=None, target_normal=None, pos_task_weight=None,
                 normal_task_weight=None, phi_limit=None, normal_offset=None):
        """Struct-of-arrays over [n, K]; a [K]-shaped array is broadcast to every frame; None keeps the current value."""
        n, K = self.n, self.K

        def prep(a, dtype, tail):
            if a is None:
                return None
            a = np.asarray(a, dtype)
            if a.shape == (K,) + tail:
                a = np.broadcast_to(a, (n, K) + tail)
            if a.shape != (n, K) + tail:
                raise SmplppError(1, "task array has shape %s, expected %s" % (a.shape, (n, K) + tail))
            return np.ascontiguousarray(a)

        args = [prep(face_idx, np.int64, ()), prep(vertex_weights, np.float32, (3,)), prep(target_pos, np.float32, (3,)),
                prep(target_normal, np.float32, (3,)), prep(pos_task_weight, np.float64, ()),
                prep(normal_task_weight, np.float64, ()), prep(phi_limit, np.float64, ()), prep(normal_offset, np.float64, ())]
        check(_lib.load().smplpp_ik_set_tasks(self._h, *[_ptr(a) for a in args], HOST))

    def setTaskList(self, ikTaskList: Dict[str, IkTask]):
        """One `std::map<std::string, IkTask>` (node.cpp:47) broadcast to every frame, in its iteration order."""
        names = sorted(ikTaskList)
        if len(names) != self.K:
            raise SmplppError(1, "expected %d tasks, got %d" % (self.K, len(names)))
        self.task_names = names
        t = [ikTaskList[k] for k in names]
        self.setTasks([x.faceIdx_ for x in t], [x.vertexWeights_ for x in t], [x.targetPos_ for x in t],
                      [x.targetNormal_ for x in t], [x.posTaskWeight_ for x in t], [x.normalTaskWeight_ for x in t],
                      [x.phiLimit_ for x in t], [x.normalOffset_ for x in t])

    def getTasks(self):
        n, K = self.n, self.K
        face = np.empty((n, K), np.int64)
        vw = np.empty((n, K, 3), np.float32)
        tang = np.empty((n, K, 3, 2), np.float32)
        apos = np.empty((n, K, 3), np.float32)
        anrm = np.empty((n, K, 3), np.float32)
        check(_lib.load().smplpp_ik_get_tasks(self._h, _ptr(face), _ptr(vw), _ptr(tang), _ptr(apos), _ptr(anrm), HOST))
        return dict(face_idx=face, vertex_weights=vw, tangents=tang, actual_pos=apos, actual_normal=anrm)

    def getStatus(self):
        """Per-frame solve outcome (bit 0: the last solve hit "LLT has numerical issue!", node.cpp:934-937; bit 1: some
        solve did since setConfig / the sequence start) — what an enqueue-only caller checks after synchronising."""
        f = np.zeros(self.n, np.int32)
        check(_lib.load().smplpp_ik_get_status(self._h, _ptr(f), HOST, None))
        return f

    # ---- configuration g_beta / g_theta (node.cpp:44-45)
    def setConfig(self, beta=None, theta=None):
        b = _np32(beta, (self.n, 10)) if beta is not None else None
        t = _np32(theta).reshape(self.n, self.theta_dim) if theta is not None else None
        check(_lib.load().smplpp_ik_set_config(self._h, _ptr(b), _ptr(t), HOST))

    def getConfig(self):
        b = np.empty((self.n, 10), np.float32)
        t = np.empty((self.n, self.theta_dim), np.float32)
        check(_lib.load().smplpp_ik_get_config(self._h, _ptr(b), _ptr(t), HOST))
        return b, (t.reshape(self.n, 25, 3) if self.theta_dim == 75 else t)

    # ---- node.cpp:750-877
    def eval(self, optimize_beta=False):
        D = self.theta_dim + 2 * self.K + (10 if optimize_beta else 0)
        e = np.empty((self.n, 4 * self.K), np.float64)
        J = np.empty((self.n, 4 * self.K, D), np.float64)
        check(_lib.load().smplpp_ik_eval(self._h, int(optimize_beta), _ptr(e), _ptr(J), HOST, None))
        return e, J

    # ---- node.cpp:704-1001 x iters
    def iterate(self, iters, enable_qp=False, optimize_beta_from=-1, min_valid=0, sync=True, stream=None):
        e2 = np.empty(self.n, np.float64) if sync else None
        check(_lib.load().smplpp_ik_iterate(self._h, int(iters), int(enable_qp), int(optimize_beta_from), int(min_valid),
                                            _ptr(e2), HOST if sync else DEVICE, stream))
        return e2

    # ---- node.cpp:1369-1407 (+ :681-700): the frame loop of solveMocapMotion without a host round trip per frame
    def solveSequence(self, target_pos, valid, warmup_iters=32, iters_per_frame=1, enable_qp=True, min_valid=0):
        """target_pos [T,n,K,3] float32, valid [T,n,K] bool -> theta [T,n,theta_dim] after every frame.  target_pos [T,K,3] with valid
        [T,K]: ONE capture for all n chains (smplpp_ik_solve_sequence_shared: the targets are repeated on the device, not here)."""
        tp = np.ascontiguousarray(target_pos, np.float32)
        vl = np.ascontiguousarray(valid, np.uint8)
        T = tp.shape[0]
        shared = tp.ndim == 3
        if shared:
            assert tp.shape == (T, self.K, 3) and vl.shape == (T, self.K)
        else:
            assert tp.shape == (T, self.n, self.K, 3) and vl.shape == (T, self.n, self.K)
        out = np.empty((T, self.n, self.theta_dim), np.float32)
        fn = _lib.load().smplpp_ik_solve_sequence_shared if shared else _lib.load().smplpp_ik_solve_sequence
        check(fn(self._h, T, _ptr(tp), _ptr(vl), int(warmup_iters), int(iters_per_frame), int(enable_qp), int(min_valid), _ptr(out), HOST, None))
        return out

    def getVertices(self):
        v = np.empty((self.n, self.smpl.vertex_num, 3), np.float32)
        check(_lib.load().smplpp_ik_get_vertices(self._h, _ptr(v), HOST, None))
        return v


def reference_task_faces(K=6):
    """The reference's four end-effector tasks (node/node.cpp:538-550: LeftFoot f5925, LeftHand f2581, RightFoot f12812,
    RightHand f9469) plus HeadTop f7324 and Chest f6842 from its mocap table (:455,:459), in std::map order."""
    table = {"LeftHand": 2581, "RightHand": 9469, "LeftFoot": 5925, "RightFoot": 12812, "HeadTop": 7324, "Chest": 6842}
    names = sorted(list(table)[:K])
    return names, np.array([table[k] for k in names], np.int64)
