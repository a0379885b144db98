"""TEST INFRASTRUCTURE ONLY — ctypes binding of oracle/liboracle.so (plain-C restatement, oracle/smpl_oracle.c).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

_f = C.POINTER(C.c_float)
_d = C.POINTER(C.c_double)
_i64 = C.POINTER(C.c_int64)
_i32 = C.POINTER(C.c_int32)


class _Tasks(C.Structure):
    _fields_ = [
        ("K", C.c_int64),
        ("face_idx", _i64),
        ("vertex_weights", _f),
        ("tangents", _f),
        ("target_pos", _f),
        ("target_normal", _f),
        ("pos_task_weight", _d),
        ("normal_task_weight", _d),
        ("phi_limit", _d),
        ("normal_offset", _d),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "smpl_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.oracle_model_create.restype = C.c_void_p
        L.oracle_model_create.argtypes = [C.c_int64, C.c_int64, _f, _f, _f, _f, _f, _i64, _i32]
        L.oracle_model_destroy.argtypes = [C.c_void_p]
        L.oracle_model_set_adjacency.argtypes = [C.c_void_p, C.c_int64, C.c_int64, _i64]
        L.oracle_model_get_adjacency.restype = C.c_int64
        L.oracle_model_get_adjacency.argtypes = [C.c_void_p, C.c_int64, C.c_int64, _i64, _f]
        L.oracle_rodrigues.argtypes = [C.c_int64, _f, _f]
        L.oracle_blend_shape.argtypes = [C.c_int64, C.c_int64, _f, _f, _f, _f, _f, _f, _f]
        L.oracle_joint_regression.argtypes = [C.c_int64, C.c_int64, _f, _f, _f, _f, _f, _f]
        L.oracle_world_transformation.argtypes = [C.c_int64, _i64, _f, _f, _f]
        L.oracle_lbs.argtypes = [C.c_int64, C.c_int64, _f, _f, _f, _f, _f]
        L.oracle_fk.argtypes = [C.c_void_p, C.c_int64, _f, _f, _f, _f, _f, _f, _f, C.c_int]
        L.oracle_max_threads.restype = C.c_int
        L.oracle_face_normal.argtypes = [C.c_void_p, _f, C.c_int64, _f]
        L.oracle_vertex_normal.argtypes = [C.c_void_p, _f, C.c_int64, _f]
        L.oracle_triangle_vertex_weights.argtypes = [_f, _f, _f]
        L.oracle_closest_points.argtypes = [C.c_void_p, _f, C.c_int64, _f, _i64, _f, _f]
        L.oracle_winding_numbers.argtypes = [C.c_void_p, _f, C.c_int64, _f, _d]
        L.oracle_ik_eval.argtypes = [C.c_void_p, _f, _f, C.POINTER(_Tasks), C.c_int, _f, _f, _d, _d, _f]
        L.oracle_normal_equations.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int64, _d, _d, _f, _d, _d]
        L.oracle_llt_solve.argtypes = [C.c_int64, _d, _d, _d]
        L.oracle_box_qp.argtypes = [C.c_int64, _d, _d, _d, _d, _d]
        L.oracle_ik_solve.argtypes = [C.c_void_p, _f, _f, C.POINTER(_Tasks), C.c_int, C.c_int, C.c_int, _d]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, np.float32)
    return a.reshape(shape) if shape is not None else a


class TaskSet:
    """Struct-of-arrays mirror of a std::map<std::string, IkTask> in iteration order (node/node.cpp:47)."""

    def __init__(self, face_idx, target_pos, target_normal=None, pos_task_weight=None, normal_task_weight=None,
                 phi_limit=None, normal_offset=None, vertex_weights=None):
        K = len(face_idx)
        self.K = K
        self.face_idx = np.ascontiguousarray(face_idx, np.int64).copy()
        self.target_pos = _f32(target_pos, (K, 3)).copy()
        self.target_normal = _f32(np.tile([0, 0, 1.0], (K, 1)) if target_normal is None else target_normal, (K, 3)).copy()
        d = lambda v, dflt: np.ascontiguousarray(np.full(K, dflt) if v is None else v, np.float64).reshape(K).copy()
        self.pos_task_weight = d(pos_task_weight, 1.0)  # IkTask.h:58
        self.normal_task_weight = d(normal_task_weight, 1.0)  # :61
        self.phi_limit = d(phi_limit, 0.04)  # :64
        self.normal_offset = d(normal_offset, 0.0)  # :67
        self.vertex_weights = _f32(np.full((K, 3), 1.0 / 3.0) if vertex_weights is None else vertex_weights, (K, 3)).copy()
        self.tangents = np.zeros((K, 3, 2), np.float32)

    def c_struct(self):
        return _Tasks(self.K, _p(self.face_idx, _i64), _p(self.vertex_weights, _f), _p(self.tangents, _f),
                      _p(self.target_pos, _f), _p(self.target_normal, _f), _p(self.pos_task_weight, _d),
                      _p(self.normal_task_weight, _d), _p(self.phi_limit, _d), _p(self.normal_offset, _d))

    def copy(self):
        return TaskSet(self.face_idx, self.target_pos, self.target_normal, self.pos_task_weight, self.normal_task_weight,
                       self.phi_limit, self.normal_offset, self.vertex_weights)


class OracleModel:
    def __init__(self, model):
        L = lib()
        self.m = {k: np.ascontiguousarray(v) for k, v in model.items()}
        self.V = self.m["vertices_template"].shape[0]
        self.F = self.m["face_indices"].shape[0]
        self.h = L.oracle_model_create(
            self.V, self.F, _p(self.m["vertices_template"], _f), _p(self.m["shape_blend_shapes"], _f),
            _p(self.m["pose_blend_shapes"], _f), _p(self.m["joint_regressor"], _f), _p(self.m["weights"], _f),
            _p(self.m["kinematic_tree"], _i64), _p(self.m["face_indices"], _i32))

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_model_destroy(self.h)
            self.h = None

    def set_adjacency(self, v, faces):
        faces = np.ascontiguousarray(faces, np.int64)
        if lib().oracle_model_set_adjacency(self.h, v, len(faces), _p(faces, _i64)):
            raise ValueError("adjacency size mismatch for vertex %d" % v)

    def get_adjacency(self, v):
        faces = np.empty(256, np.int64)
        w = np.empty(256, np.float32)
        n = lib().oracle_model_get_adjacency(self.h, v, 256, _p(faces, _i64), _p(w, _f))
        return faces[:n].copy(), w[:n].copy()

    def fk(self, beta, theta, want=("verts", "joints", "xforms", "rest", "poserot"), threads=0):
        beta = _f32(beta)
        theta = _f32(theta)
        n = beta.shape[0]
        out = {
            "verts": np.empty((n, self.V, 3), np.float32) if "verts" in want else None,
            "joints": np.empty((n, 24, 3), np.float32) if "joints" in want else None,
            "xforms": np.empty((n, 24, 4, 4), np.float32) if "xforms" in want else None,
            "rest": np.empty((n, self.V, 3), np.float32) if "rest" in want else None,
            "poserot": np.empty((n, 24, 3, 3), np.float32) if "poserot" in want else None,
        }
        lib().oracle_fk(self.h, n, _p(beta, _f), _p(theta, _f), _p(out["verts"], _f), _p(out["joints"], _f),
                        _p(out["xforms"], _f), _p(out["rest"], _f), _p(out["poserot"], _f), threads)
        return {k: v for k, v in out.items() if v is not None}

    def face_normal(self, verts, face):
        verts = _f32(verts)
        n = np.empty(3, np.float32)
        lib().oracle_face_normal(self.h, _p(verts, _f), face, _p(n, _f))
        return n

    def vertex_normal(self, verts, vertex):
        verts = _f32(verts)
        n = np.empty(3, np.float32)
        lib().oracle_vertex_normal(self.h, _p(verts, _f), vertex, _p(n, _f))
        return n

    def winding_numbers(self, verts, points):
        """igl::winding_number (node/node.cpp:1052) at points [K,3] for one frame's vertices."""
        verts = _f32(verts)
        points = _f32(points).reshape(-1, 3)
        w = np.empty(points.shape[0], np.float64)
        lib().oracle_winding_numbers(self.h, _p(verts, _f), points.shape[0], _p(points, _f), _p(w, _d))
        return w

    def closest_points(self, verts, points):
        verts = _f32(verts)
        points = _f32(points).reshape(-1, 3)
        K = points.shape[0]
        face = np.empty(K, np.int64)
        closest = np.empty((K, 3), np.float32)
        sq = np.empty(K, np.float32)
        lib().oracle_closest_points(self.h, _p(verts, _f), K, _p(points, _f), _p(face, _i64), _p(closest, _f), _p(sq, _f))
        return face, closest, sq

    def ik_eval(self, beta, theta, tasks: TaskSet, optimize_beta=False, want_verts=False):
        beta = _f32(beta, (10,))
        theta = _f32(theta, (25, 3))
        K = tasks.K
        D = 75 + 2 * K + (10 if optimize_beta else 0)
        e = np.empty(4 * K, np.float64)
        J = np.empty((4 * K, D), np.float64)
        apos = np.empty((K, 3), np.float32)
        anrm = np.empty((K, 3), np.float32)
        verts = np.empty((self.V, 3), np.float32) if want_verts else None
        ts = tasks.c_struct()
        lib().oracle_ik_eval(self.h, _p(beta, _f), _p(theta, _f), C.byref(ts), int(optimize_beta), _p(apos, _f),
                             _p(anrm, _f), _p(e, _d), _p(J, _d), _p(verts, _f))
        return dict(e=e, J=J, actual_pos=apos, actual_normal=anrm, verts=verts)

    def ik_solve(self, beta, theta, tasks: TaskSet, iters, enable_qp=False, optimize_beta_from=-1):
        beta = _f32(beta, (10,)).copy()
        theta = _f32(theta, (25, 3)).copy()
        e2 = C.c_double(0.0)
        ts = tasks.c_struct()
        rc = lib().oracle_ik_solve(self.h, _p(beta, _f), _p(theta, _f), C.byref(ts), iters, int(enable_qp),
                                   optimize_beta_from, C.byref(e2))
        if rc:
            raise RuntimeError("LLT has numerical issue!")  # node/node.cpp:936
        return beta, theta, e2.value


def rodrigues(theta):
    theta = _f32(theta).reshape(-1, 24, 3)
    out = np.empty(theta.shape + (3,), np.float32)
    lib().oracle_rodrigues(theta.shape[0], _p(theta, _f), _p(out, _f))
    return out


def blend_shape(beta, theta, S, P):
    beta, theta, S, P = _f32(beta), _f32(theta), _f32(S), _f32(P)
    n, V = beta.shape[0], S.shape[0]
    bs = np.empty((n, V, 3), np.float32)
    bp = np.empty((n, V, 3), np.float32)
    rot = np.empty((n, 24, 3, 3), np.float32)
    lib().oracle_blend_shape(V, n, _p(beta, _f), _p(theta, _f), _p(S, _f), _p(P, _f), _p(bs, _f), _p(bp, _f), _p(rot, _f))
    return bs, bp, rot


def joint_regression(T, Jreg, bs, bp):
    T, Jreg, bs, bp = _f32(T), _f32(Jreg), _f32(bs), _f32(bp)
    n, V = bs.shape[0], T.shape[0]
    rest = np.empty((n, V, 3), np.float32)
    joints = np.empty((n, 24, 3), np.float32)
    lib().oracle_joint_regression(V, n, _p(T, _f), _p(Jreg, _f), _p(bs, _f), _p(bp, _f), _p(rest, _f), _p(joints, _f))
    return rest, joints


def world_transformation(kintree, joints, pose_rot):
    kintree = np.ascontiguousarray(kintree, np.int64)
    joints, pose_rot = _f32(joints), _f32(pose_rot)
    n = joints.shape[0]
    out = np.empty((n, 24, 4, 4), np.float32)
    lib().oracle_world_transformation(n, _p(kintree, _i64), _p(joints, _f), _p(pose_rot, _f), _p(out, _f))
    return out


def lbs(W, rest, xforms, root_pos=None):
    W, rest, xforms = _f32(W), _f32(rest), _f32(xforms)
    n, V = rest.shape[0], W.shape[0]
    root = _f32(root_pos).reshape(n, 3) if root_pos is not None else None
    out = np.empty((n, V, 3), np.float32)
    lib().oracle_lbs(V, n, _p(W, _f), _p(rest, _f), _p(xforms, _f), _p(root, _f), _p(out, _f))
    return out


def triangle_vertex_weights(pos, tri):
    pos, tri = _f32(pos, (3,)), _f32(tri, (3, 3))
    w = np.empty(3, np.float32)
    lib().oracle_triangle_vertex_weights(_p(pos, _f), _p(tri, _f), _p(w, _f))
    return w


def normal_equations(e, J, theta_dim, phi_dim, beta_dim, vposer_theta=None):
    e = np.ascontiguousarray(e, np.float64)
    J = np.ascontiguousarray(J, np.float64)
    D = theta_dim + phi_dim + beta_dim
    A = np.empty((D, D), np.float64)
    b = np.empty(D, np.float64)
    vt = _f32(vposer_theta) if vposer_theta is not None else None
    lib().oracle_normal_equations(J.shape[0], theta_dim, phi_dim, beta_dim, _p(e, _d), _p(J, _d), _p(vt, _f), _p(A, _d), _p(b, _d))
    return A, b


def llt_solve(A, b):
    A = np.ascontiguousarray(A, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    x = np.empty_like(b)
    if lib().oracle_llt_solve(len(b), _p(A, _d), _p(b, _d), _p(x, _d)):
        raise RuntimeError("LLT has numerical issue!")
    return x


def box_qp(A, b, lo, hi):
    A = np.ascontiguousarray(A, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    lo = np.ascontiguousarray(np.clip(lo, -1e30, 1e30), np.float64)
    hi = np.ascontiguousarray(np.clip(hi, -1e30, 1e30), np.float64)
    x = np.empty_like(b)
    if lib().oracle_box_qp(len(b), _p(A, _d), _p(b, _d), _p(lo, _d), _p(hi, _d), _p(x, _d)):
        raise RuntimeError("box QP did not converge")
    return x
