"""TEST INFRASTRUCTURE ONLY — ctypes binding of oracle/_ref/libsmplpp_ref.so (the reference's own compiled FK
stages + oracle/ref_driver.cpp).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this; the product package smplpp_amd/ never does.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libsmplpp_ref.so")

_f = C.POINTER(C.c_float)
_d = C.POINTER(C.c_double)
_i64 = C.POINTER(C.c_int64)
_i32 = C.POINTER(C.c_int32)


def available() -> bool:
    return os.path.exists(LIB_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (loads libtorch/libc10 so the rpath-less case also resolves)

        L = C.CDLL(LIB_PATH)
        L.ref_last_error.restype = C.c_char_p
        L.ref_model_create.restype = C.c_void_p
        L.ref_model_create.argtypes = [C.c_int64, C.c_int64, _f, _f, _f, _f, _f, _i64, _i32]
        L.ref_model_destroy.argtypes = [C.c_void_p]
        L.ref_fk.argtypes = [C.c_void_p, C.c_int64, _f, _f, _f, _f, _f, _f, _f]
        L.ref_fk_launch_only.argtypes = [C.c_void_p, C.c_int64, _f, _f]
        L.ref_adjacent_faces.restype = C.c_int64
        L.ref_adjacent_faces.argtypes = [C.c_void_p, C.c_int64, C.c_int64, _i64, _f]
        L.ref_ik_eval.argtypes = [C.c_void_p, _f, _f, C.c_int64, _i64, _f, _f, _d, _d, _d, _d, C.c_int,
                                  _f, _f, _f, _f, _d, _d]
        L.ref_set_num_threads.argtypes = [C.c_int]
        L.ref_get_num_threads.restype = C.c_int
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


class RefModel:
    def __init__(self, model):
        L = lib()
        self.m = {k: np.ascontiguousarray(v) for k, v in model.items()}
        self.V = self.m["vertices_template"].shape[0]
        self.F = self.m["face_indices"].shape[0]
        self.h = L.ref_model_create(
            self.V, self.F, _p(self.m["vertices_template"], _f), _p(self.m["shape_blend_shapes"], _f),
            _p(self.m["pose_blend_shapes"], _f), _p(self.m["joint_regressor"], _f), _p(self.m["weights"], _f),
            _p(self.m["kinematic_tree"], _i64), _p(self.m["face_indices"], _i32))
        if not self.h:
            raise RuntimeError(L.ref_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_model_destroy(self.h)
            self.h = None

    def fk(self, beta, theta, want=("verts", "joints", "xforms", "rest", "poserot")):
        L = lib()
        beta = np.ascontiguousarray(beta, np.float32)
        theta = np.ascontiguousarray(theta, np.float32)
        n = beta.shape[0]
        out = {
            "verts": np.empty((n, self.V, 3), np.float32) if "verts" in want else None,
            "joints": np.empty((n, 24, 3), np.float32) if "joints" in want else None,
            "xforms": np.empty((n, 24, 4, 4), np.float32) if "xforms" in want else None,
            "rest": np.empty((n, self.V, 3), np.float32) if "rest" in want else None,
            "poserot": np.empty((n, 24, 3, 3), np.float32) if "poserot" in want else None,
        }
        rc = L.ref_fk(self.h, n, _p(beta, _f), _p(theta, _f), _p(out["verts"], _f), _p(out["joints"], _f),
                      _p(out["xforms"], _f), _p(out["rest"], _f), _p(out["poserot"], _f))
        if rc:
            raise RuntimeError(L.ref_last_error().decode())
        return {k: v for k, v in out.items() if v is not None}

    def fk_launch_only(self, beta, theta):
        L = lib()
        rc = L.ref_fk_launch_only(self.h, beta.shape[0], _p(beta, _f), _p(theta, _f))
        if rc:
            raise RuntimeError(L.ref_last_error().decode())

    def adjacent_faces(self, v):
        L = lib()
        faces = np.empty(64, np.int64)
        w = np.empty(64, np.float32)
        n = L.ref_adjacent_faces(self.h, v, 64, _p(faces, _i64), _p(w, _f))
        return faces[:n].copy(), w[:n].copy()

    def ik_eval(self, beta, theta, face_idx, target_pos, target_normal, pos_w, normal_w, phi_limit, normal_offset,
                vertex_weights, optimize_beta=False):
        """node/node.cpp:704-877 for one frame. Returns dict(e, J, vertex_weights, tangents, actual_pos, actual_normal)."""
        L = lib()
        K = len(face_idx)
        D = 75 + 2 * K + (10 if optimize_beta else 0)
        beta = np.ascontiguousarray(beta, np.float32).reshape(10)
        theta = np.ascontiguousarray(theta, np.float32).reshape(25, 3)
        face_idx = np.ascontiguousarray(face_idx, np.int64)
        tp = np.ascontiguousarray(target_pos, np.float32).reshape(K, 3)
        tn = np.ascontiguousarray(target_normal, np.float32).reshape(K, 3)
        pw = np.ascontiguousarray(pos_w, np.float64).reshape(K)
        nw = np.ascontiguousarray(normal_w, np.float64).reshape(K)
        pl = np.ascontiguousarray(phi_limit, np.float64).reshape(K)
        no = np.ascontiguousarray(normal_offset, np.float64).reshape(K)
        vw = np.array(vertex_weights, np.float32).reshape(K, 3).copy()
        tang = np.empty((K, 3, 2), np.float32)
        apos = np.empty((K, 3), np.float32)
        anrm = np.empty((K, 3), np.float32)
        e = np.empty(4 * K, np.float64)
        J = np.empty((4 * K, D), np.float64)
        rc = L.ref_ik_eval(self.h, _p(beta, _f), _p(theta, _f), K, _p(face_idx, _i64), _p(tp, _f), _p(tn, _f),
                           _p(pw, _d), _p(nw, _d), _p(pl, _d), _p(no, _d), int(optimize_beta),
                           _p(vw, _f), _p(tang, _f), _p(apos, _f), _p(anrm, _f), _p(e, _d), _p(J, _d))
        if rc:
            raise RuntimeError(L.ref_last_error().decode())
        return dict(e=e, J=J, vertex_weights=vw, tangents=tang, actual_pos=apos, actual_normal=anrm)
