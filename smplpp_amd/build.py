"""Build libsmplpp_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsmplpp_hip.so")
ARCH = "gfx950"
# skin_h.hip: accumulators in arch VGPRs (its VALU epilogue reads them; the A operand takes the AGPRs).
# skin_p.hip places its VALU work by hand in MFMA shadows: SLP-packing adjacent f32 FMAs into v_pk_fma_f32 (+ the v_mov
# shuffles that feeds them) is an anti-lever beside MFMAs (cdna_hip_programming.md, per-instruction constants).
PER_FILE_FLAGS = {"skin_p.hip": ["-fno-slp-vectorize"], "skin_b.hip": ["-fno-slp-vectorize"], "skin_e.hip": ["-fno-slp-vectorize"], "skin_h.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"], "fk.hip": ["-fno-slp-vectorize"], "ik.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libsmplpp_hip.so)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    objs = []
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    flags = ["-O3", "--offload-arch=" + ARCH, "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
        objs.append(obj)
        if not force and os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in [src] + hdrs):
            continue
        extra = PER_FILE_FLAGS.get(os.path.basename(src), [])
        cmd = [hipcc] + flags + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out))
        if verbose and out.strip():
            print(out, file=sys.stderr)
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
