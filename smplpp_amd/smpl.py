"""Host-side mirror of the reference's `smplpp::SMPL` façade (include/smplpp/SMPL.h:140-270) over the C ABI.

Same method names and argument meaning as the reference class; tensors are numpy arrays (host: the call stages
and synchronises) or torch tensors on the MI355X (device: the call only enqueues on torch's current stream).
PyTorch is used for device memory and streams only — every number is produced by libsmplpp_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib, model_io
from ._lib import DEVICE, HOST, SmplppError, check

try:  # torch is plumbing (device buffers, streams); the package works with numpy alone
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_torch(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor)


def _np32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a.reshape(shape) if shape is not None else a


def _ptr(a):
    """Address of a numpy array / torch tensor / None."""
    if a is None:
        return None
    if _is_torch(a):
        return a.data_ptr()
    return a.ctypes.data


def _stream():
    if torch is not None and torch.cuda.is_available():
        return torch.cuda.current_stream().cuda_stream
    return None


def parse_device(device) -> int:
    """Reference: `torch::Device` with an explicit index (src/SMPL.cpp:289-297); "CUDA" selects the GPU engine
    (node/node.cpp:360-371).  There is no CPU engine here."""
    if isinstance(device, int):
        return device
    s = str(device).lower()
    if s.startswith("cpu"):
        raise SmplppError(1, "smplpp_amd has no CPU engine: use device 'cuda:<i>' / 'hip:<i>' (MI355X)")
    if ":" not in s:
        raise SmplppError(1, "Failed to fetch device index!")  # src/SMPL.cpp:295
    return int(s.split(":")[1])


class SMPL:
    def __init__(self):
        self._h = None
        self._device = 0
        self._path = None
        self._model = None
        self._out = {}
        self._n = 0

    # ---- setters / init (SMPL.h:241-246)
    def setDevice(self, device):
        self._device = parse_device(device)

    def getDevice(self):
        return "cuda:%d" % self._device

    def setModelPath(self, modelPath: str):
        self._path = modelPath

    def init(self, model: Optional[dict] = None):
        """SMPL::init (src/SMPL.cpp:560-643).  `model` (the seven arrays of scripts/preprocess.py:98-117) may be
        passed directly instead of a path — needed here because the real parameter files are license-gated."""
        if model is None:
            if self._path is None:
                raise SmplppError(1, "Cannot initialize a SMPL model!")
            model = model_io.load_model(self._path)
        m = model_io._normalise(model)
        self._model = m
        L = _lib.load()
        _lib.require_gpu()
        if self._h:
            check(L.smplpp_model_destroy(self._h))
            self._h = None
        h = C.c_void_p()
        check(L.smplpp_model_create(
            m["vertices_template"].shape[0], m["face_indices"].shape[0], _ptr(m["vertices_template"]),
            _ptr(m["shape_blend_shapes"]), _ptr(m["pose_blend_shapes"]), _ptr(m["joint_regressor"]), _ptr(m["weights"]),
            _ptr(m["kinematic_tree"]), _ptr(m["face_indices"]), self._device, C.byref(h)))
        self._h = h
        self.vertex_num = m["vertices_template"].shape[0]
        self.face_num = m["face_indices"].shape[0]

    def __del__(self):
        try:
            if self._h:
                _lib.load().smplpp_model_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def handle(self):
        if not self._h:
            raise SmplppError(4, "Cannot launch a SMPL model!")  # src/SMPL.cpp:676
        return self._h

    def info(self):
        V, F, w, d = C.c_int64(), C.c_int64(), C.c_int(), C.c_int()
        check(_lib.load().smplpp_model_info(self.handle, C.byref(V), C.byref(F), C.byref(w), C.byref(d)))
        return dict(vertex_num=V.value, face_num=F.value, weights_per_vertex=w.value, device=d.value)

    # ---- measurement hook
    def profileEnable(self, enable=True):
        check(_lib.load().smplpp_profile_enable(self.handle, int(enable)))

    def profileRead(self):
        """(launches, mean fused-kernel duration in ms) since the last read; HIP events on the launch stream."""
        n, ms = C.c_int64(), C.c_double()
        check(_lib.load().smplpp_profile_read(self.handle, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    # ---- launch (SMPL.h:268, src/SMPL.cpp:671-737)
    def launch(self, beta, theta, want=("verts", "joints", "xforms", "rest"), out=None):
        """beta [N,10], theta [N,25,3] (row 0 = root translation).  Outputs are kept for the getters.
        `out` (optional dict of preallocated arrays/tensors keyed like `want`) avoids per-call allocation."""
        V = self.vertex_num
        L = _lib.load()
        if _is_torch(beta) != _is_torch(theta):
            raise SmplppError(1, "Cannot launch a SMPL model!")
        if _is_torch(beta):
            if not (beta.is_cuda and theta.is_cuda and beta.dtype == torch.float32 and theta.dtype == torch.float32):
                raise SmplppError(1, "Cannot launch a SMPL model!")
            beta, theta = beta.contiguous(), theta.contiguous()
            n = beta.shape[0]
            if tuple(beta.shape) != (n, 10) or tuple(theta.shape) != (n, 25, 3):
                raise SmplppError(1, "Cannot launch a SMPL model!")
            mk = lambda *s: torch.empty(s, dtype=torch.float32, device=beta.device)
            space = DEVICE
        else:
            beta, theta = _np32(beta), _np32(theta)
            n = beta.shape[0]
            if beta.shape != (n, 10) or theta.shape != (n, 25, 3):
                raise SmplppError(1, "Cannot launch a SMPL model!")
            mk = lambda *s: np.empty(s, np.float32)
            space = HOST
        pre = out or {}
        out = {
            "verts": pre.get("verts", mk(n, V, 3) if "verts" in want else None),
            "joints": pre.get("joints", mk(n, 24, 3) if "joints" in want else None),
            "xforms": pre.get("xforms", mk(n, 24, 4, 4) if "xforms" in want else None),
            "rest": pre.get("rest", mk(n, V, 3) if "rest" in want else None),
        }
        check(L.smplpp_fk(self.handle, n, _ptr(beta), _ptr(theta), _ptr(out["verts"]), _ptr(out["joints"]),
                          _ptr(out["xforms"]), _ptr(out["rest"]), space, _stream() if space == DEVICE else None))
        self._out, self._n, self._theta = out, n, theta
        return out

    def launchStatus(self):
        """Status word of the launches since the last read (synchronises the current stream): bit 0 = an operand left the
        input range of the default fused kernel (include/smplpp_hip.h, smplpp_fk_status).  Host-space launches raise instead."""
        bits = C.c_int(0)
        check(_lib.load().smplpp_fk_status(self.handle, C.byref(bits), _stream()))
        return bits.value

    def _need(self, key):
        if self._out.get(key) is None:
            raise SmplppError(4, "Failed to get vertices of new pose!")  # src/LinearBlendSkinning.cpp:413
        return self._out[key]

    # ---- getters (SMPL.h:248-262)
    def getVertex(self):
        v = self._need("verts")
        return v.clone() if _is_torch(v) else v.copy()  # :492-506 returns a clone

    def getRestShape(self):
        v = self._need("rest")
        return v.clone() if _is_torch(v) else v.copy()

    def getRestJoint(self):
        v = self._need("joints")
        return v.clone() if _is_torch(v) else v.copy()

    def getTransformation(self):
        v = self._need("xforms")
        return v.clone() if _is_torch(v) else v.copy()

    def getFaceIndex(self):
        return self._model["face_indices"].copy()  # [F,3] int32, 1-based (src/SMPL.cpp:418-433)

    def getFaceIndexRaw(self, idx):
        return self._model["face_indices"][idx]  # :435-438

    def getVertexRaw(self, idx):
        return self._need("verts")[0, idx]  # batch 0 only (src/LinearBlendSkinning.cpp:419-427)

    def getAdjacentFaces(self, idx):
        """{face id: weight} like the reference's unordered_map (src/SMPL.cpp:537-540)."""
        faces = (C.c_int64 * 64)()
        w = (C.c_float * 64)()
        cnt = C.c_int64()
        check(_lib.load().smplpp_adjacent_faces(self.handle, int(idx), 64, faces, w, C.byref(cnt)))
        return {int(faces[i]): float(w[i]) for i in range(min(cnt.value, 64))}

    def _normals(self, ids, vertex, frame=None):
        verts = self._need("verts")
        single = np.isscalar(ids)
        ids_np = np.ascontiguousarray(np.atleast_1d(ids), np.int64)
        L = _lib.load()
        fn = L.smplpp_vertex_normals if vertex else L.smplpp_face_normals
        if _is_torch(verts):
            v = verts if frame is None else verts[frame:frame + 1]
            idt = torch.from_numpy(ids_np).to(verts.device)
            out = torch.empty((v.shape[0], len(ids_np), 3), dtype=torch.float32, device=verts.device)
            check(fn(self.handle, v.shape[0], _ptr(v), len(ids_np), _ptr(idt), _ptr(out), DEVICE, _stream()))
        else:
            v = verts if frame is None else verts[frame:frame + 1]
            out = np.empty((v.shape[0], len(ids_np), 3), np.float32)
            check(fn(self.handle, v.shape[0], _ptr(v), len(ids_np), _ptr(ids_np), _ptr(out), HOST, None))
        if frame is not None:
            out = out[0]
            return out[0] if single else out
        return out

    def calcNormal(self, faceIdx):
        """SMPL::calcNormal (src/SMPL.cpp:518-525): batch 0, like the reference."""
        return self._normals(faceIdx, False, frame=0)

    def calcVertexNormal(self, idx):
        """SMPL::calcVertexNormal (src/SMPL.cpp:527-535): batch 0."""
        return self._normals(idx, True, frame=0)

    def calcNormalBatch(self, faceIds):
        return self._normals(faceIds, False)

    def calcVertexNormalBatch(self, vertexIds):
        return self._normals(vertexIds, True)

    def calcMeshVertexNormals(self):
        """SMPL::calcVertexNormal (src/SMPL.cpp:527-535) for every vertex of every frame of the last launch: [N,V,3]."""
        verts = self._need("verts")
        L = _lib.load()
        if _is_torch(verts):
            out = torch.empty_like(verts)
            check(L.smplpp_mesh_vertex_normals(self.handle, verts.shape[0], _ptr(verts), _ptr(out), DEVICE, _stream()))
        else:
            out = np.empty_like(verts)
            check(L.smplpp_mesh_vertex_normals(self.handle, verts.shape[0], _ptr(verts), _ptr(out), HOST, None))
        return out

    def calcSweepGrid(self, frame=0):
        """The sweep grid of node/node.cpp:1023-1073 for one frame of the last launch: dict(grid_min [3], grid_num [3],
        winding [cells], inside [cells] bool, grid_idx [cells,3] int32 in the reference's cell order, positions = 0.025 *
        grid_idx). `inside` marks the cells the reference enters into g_sweepGridList (winding number > 0.5)."""
        verts = self._need("verts")
        v = verts[frame]
        if _is_torch(v):
            v = v.contiguous()
            space, st = DEVICE, _stream()
        else:
            v = np.ascontiguousarray(v)
            space, st = HOST, None
        L = _lib.load()
        gmin = np.zeros(3, np.int32)
        gnum = np.zeros(3, np.int32)
        cells = C.c_int64(0)
        check(L.smplpp_sweep_grid(self.handle, _ptr(v), _ptr(gmin), _ptr(gnum), 0, None, None, C.byref(cells), space, st))
        n = int(cells.value)
        if space == DEVICE:
            w = torch.empty(n, dtype=torch.float32, device=v.device)
            ins = torch.empty(n, dtype=torch.uint8, device=v.device)
        else:
            w = np.empty(n, np.float32)
            ins = np.empty(n, np.uint8)
        check(L.smplpp_sweep_grid(self.handle, _ptr(v), _ptr(gmin), _ptr(gnum), n, _ptr(w), _ptr(ins), C.byref(cells), space, st))
        if space == DEVICE:
            torch.cuda.synchronize()
            w, ins = w.cpu().numpy(), ins.cpu().numpy()
        ix, iy, iz = np.meshgrid(*[np.arange(gmin[a], gmin[a] + gnum[a], dtype=np.int32) for a in range(3)], indexing="ij")
        gidx = np.stack([ix.reshape(-1), iy.reshape(-1), iz.reshape(-1)], axis=1)
        return dict(grid_min=gmin, grid_num=gnum, winding=w, inside=ins.astype(bool), grid_idx=gidx)

    def closestPoints(self, points):
        """igl::point_mesh_squared_distance as used at node/node.cpp:982 — points [N,K,3] vs each frame's mesh."""
        verts = self._need("verts")
        L = _lib.load()
        n = verts.shape[0]
        if _is_torch(verts):
            points = points.contiguous()
            K = points.shape[1]
            face = torch.empty((n, K), dtype=torch.int64, device=verts.device)
            closest = torch.empty((n, K, 3), dtype=torch.float32, device=verts.device)
            sq = torch.empty((n, K), dtype=torch.float32, device=verts.device)
            check(L.smplpp_closest_points(self.handle, n, _ptr(verts), K, _ptr(points), _ptr(face), _ptr(closest), _ptr(sq),
                                          DEVICE, _stream()))
        else:
            points = _np32(points).reshape(n, -1, 3)
            K = points.shape[1]
            face = np.empty((n, K), np.int64)
            closest = np.empty((n, K, 3), np.float32)
            sq = np.empty((n, K), np.float32)
            check(L.smplpp_closest_points(self.handle, n, _ptr(verts), K, _ptr(points), _ptr(face), _ptr(closest), _ptr(sq),
                                          HOST, None))
        return face, closest, sq

    def out(self, index: int, path: str):
        """SMPL::out (src/SMPL.cpp:757-790): Wavefront OBJ of frame `index` (v lines, then 1-based f lines)."""
        verts = self._need("verts")
        v = verts[index].detach().cpu().numpy() if _is_torch(verts) else verts[index]
        with open(path, "w") as f:
            for p in v:
                f.write("v %f %f %f\n" % (p[0], p[1], p[2]))
            for t in self._model["face_indices"]:
                f.write("f %d %d %d\n" % (t[0], t[1], t[2]))


# ---- stage classes' functional forms (BlendShape / JointRegression / WorldTransformation / LinearBlendSkinning)
def stage_blend_shape(beta, theta24, shape_basis, pose_basis, device=0):
    beta, theta24, S, P = _np32(beta), _np32(theta24), _np32(shape_basis), _np32(pose_basis)
    n, V = beta.shape[0], S.shape[0]
    bs, bp, rot = np.empty((n, V, 3), np.float32), np.empty((n, V, 3), np.float32), np.empty((n, 24, 3, 3), np.float32)
    check(_lib.load().smplpp_stage_blend_shape(device, V, n, _ptr(beta), _ptr(theta24), _ptr(S), _ptr(P), _ptr(bs), _ptr(bp),
                                               _ptr(rot), HOST, None))
    return bs, bp, rot


def stage_joint_regression(T, Jreg, shape_blend, pose_blend, device=0):
    T, Jreg, bs, bp = _np32(T), _np32(Jreg), _np32(shape_blend), _np32(pose_blend)
    n, V = bs.shape[0], T.shape[0]
    rest, joints = np.empty((n, V, 3), np.float32), np.empty((n, 24, 3), np.float32)
    check(_lib.load().smplpp_stage_joint_regression(device, V, n, _ptr(T), _ptr(Jreg), _ptr(bs), _ptr(bp), _ptr(rest),
                                                    _ptr(joints), HOST, None))
    return rest, joints


def stage_world_transformation(kintree, joints, pose_rot, device=0):
    kt = np.ascontiguousarray(kintree, np.int64)
    joints, pose_rot = _np32(joints), _np32(pose_rot)
    n = joints.shape[0]
    out = np.empty((n, 24, 4, 4), np.float32)
    check(_lib.load().smplpp_stage_world_transformation(device, n, _ptr(kt), _ptr(joints), _ptr(pose_rot), _ptr(out), HOST, None))
    return out


def stage_skinning(weights, rest, xforms, root_pos=None, device=0):
    W, rest, xforms = _np32(weights), _np32(rest), _np32(xforms)
    n, V = rest.shape[0], W.shape[0]
    root = _np32(root_pos).reshape(n, 3) if root_pos is not None else None
    out = np.empty((n, V, 3), np.float32)
    check(_lib.load().smplpp_stage_skinning(device, V, n, _ptr(W), _ptr(rest), _ptr(xforms), _ptr(root), _ptr(out), HOST, None))
    return out
