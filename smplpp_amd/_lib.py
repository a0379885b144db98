"""ctypes loader for libsmplpp_hip.so (the C ABI declared in include/smplpp_hip.h).

There is no CPU fallback: if the library is missing, or no MI355X is visible, every compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMPLPP_HIP_LIB") or os.path.join(HERE, "libsmplpp_hip.so")  # override: A/B of two builds on one box
HEADER = os.path.join(os.path.dirname(HERE), "include", "smplpp_hip.h")

HOST, DEVICE = 0, 1
OK = 0

f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
i64p = C.POINTER(C.c_int64)
i32p = C.POINTER(C.c_int32)
vp = C.c_void_p


class SmplppError(RuntimeError):
    """Raised for any non-zero status; mirrors smplpp::Exception (include/smplpp/toolbox/Exception.h:121-153)."""

    def __init__(self, code, msg):
        super().__init__("[smplpp_hip error %d] %s" % (code, msg))
        self.code = code


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libsmplpp_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`python smplpp_amd/build.py`. There is no CPU fallback." % LIB_PATH)
    # ONE HIP runtime per process: the PyTorch wheel bundles its own libamdhip64.so (SONAME libamdhip64.so.7, the
    # same SONAME as /opt/rocm's).  If torch is imported first the dynamic loader binds this library to the copy
    # torch already mapped; the other order would map two runtimes (and torch then reports "No HIP GPUs").
    try:
        import torch  # noqa: F401
    except Exception:  # torch is optional plumbing; a pure-C++ host uses /opt/rocm's runtime
        pass
    L = C.CDLL(LIB_PATH)
    L.smplpp_last_error.restype = C.c_char_p
    # pointers that may be host OR device addresses are declared void* (integers from tensor.data_ptr() pass through)
    sig = {
        "smplpp_device_count": [C.POINTER(C.c_int)],
        "smplpp_model_create": [C.c_int64, C.c_int64, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.POINTER(vp)],
        "smplpp_model_destroy": [vp],
        "smplpp_model_info": [vp, i64p, i64p, C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "smplpp_profile_enable": [vp, C.c_int],
        "smplpp_profile_read": [vp, i64p, f64p],
        "smplpp_fk": [vp, C.c_int64, vp, vp, vp, vp, vp, vp, C.c_int, vp],
        "smplpp_fk_status": [vp, C.POINTER(C.c_int), vp],
        "smplpp_stage_blend_shape": [C.c_int, C.c_int64, C.c_int64, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp],
        "smplpp_stage_joint_regression": [C.c_int, C.c_int64, C.c_int64, vp, vp, vp, vp, vp, vp, C.c_int, vp],
        "smplpp_stage_world_transformation": [C.c_int, C.c_int64, vp, vp, vp, vp, C.c_int, vp],
        "smplpp_stage_skinning": [C.c_int, C.c_int64, C.c_int64, vp, vp, vp, vp, vp, C.c_int, vp],
        "smplpp_face_normals": [vp, C.c_int64, vp, C.c_int64, vp, vp, C.c_int, vp],
        "smplpp_vertex_normals": [vp, C.c_int64, vp, C.c_int64, vp, vp, C.c_int, vp],
        "smplpp_closest_points": [vp, C.c_int64, vp, C.c_int64, vp, vp, vp, vp, C.c_int, vp],
        "smplpp_mesh_vertex_normals": [vp, C.c_int64, vp, vp, C.c_int, vp],
        "smplpp_sweep_grid": [vp, vp, vp, vp, C.c_int64, vp, vp, i64p, C.c_int, vp],
        "smplpp_adjacent_faces": [vp, C.c_int64, C.c_int64, i64p, f32p, i64p],
        "smplpp_ik_create": [vp, C.c_int64, C.c_int64, vp, C.POINTER(vp)],
        "smplpp_ik_destroy": [vp],
        "smplpp_ik_set_frame_base": [vp, C.c_int64],
        "smplpp_ik_set_tasks": [vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int],
        "smplpp_ik_set_config": [vp, vp, vp, C.c_int],
        "smplpp_ik_get_config": [vp, vp, vp, C.c_int],
        "smplpp_ik_get_tasks": [vp, vp, vp, vp, vp, vp, C.c_int],
        "smplpp_ik_eval": [vp, C.c_int, vp, vp, C.c_int, vp],
        "smplpp_ik_iterate": [vp, C.c_int, C.c_int, C.c_int, C.c_int64, vp, C.c_int, vp],
        "smplpp_ik_solve_sequence": [vp, C.c_int64, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int64, vp, C.c_int, vp],
        "smplpp_ik_solve_sequence_shared": [vp, C.c_int64, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int64, vp, C.c_int, vp],
        "smplpp_ik_get_vertices": [vp, vp, C.c_int, vp],
        "smplpp_ik_get_status": [vp, vp, C.c_int, vp],
        "smplpp_gather": [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int64, vp],
        "smplpp_gather_to_root": [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int64, vp],
        "smplpp_gather_offsets": [i64p, C.c_int, C.c_int64, i64p],
        "smplpp_gather_selfcheck": [vp, C.c_int, vp, vp, C.c_int64, vp],
        "smplpp_vposer_create": [C.c_int, vp, vp, vp, vp, vp, vp, C.POINTER(vp)],
        "smplpp_vposer_destroy": [vp],
        "smplpp_vposer_forward": [vp, C.c_int64, vp, vp, vp, C.c_int, vp],
        "smplpp_vposer_forward_at": [vp, C.c_int64, C.c_int64, vp, vp, vp, C.c_int, vp],
        "smplpp_rotmat_to_axis_angle": [C.c_int, C.c_int64, vp, vp, C.c_int, vp],
    }
    for name, argtypes in sig.items():
        fn = getattr(L, name, None)
        if fn is None:  # tests/test_abi.py asserts that every declared symbol is exported
            continue
        fn.argtypes = argtypes
        fn.restype = C.c_int
    _lib = L
    return L


def declared_symbols():
    """Function names declared in include/smplpp_hip.h."""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(smplpp_[a-z0-9_]+)\s*\(", txt)))


def check(rc):
    if rc != OK:
        raise SmplppError(rc, load().smplpp_last_error().decode(errors="replace"))


def device_count() -> int:
    n = C.c_int(0)
    rc = load().smplpp_device_count(C.byref(n))
    return n.value if rc == OK else 0


def require_gpu():
    n = C.c_int(0)
    check(load().smplpp_device_count(C.byref(n)))
    return n.value
