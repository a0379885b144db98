"""The header-only C++ shim (include/smplpp/SMPL.h, IkTask.h: the reference's class names over the C ABI) compiles
with plain g++ and links libsmplpp_hip.so; without a GPU it fails loudly with smplpp::Exception, with one it runs FK
and an IK iteration on a small model read from the reference's .json schema."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shim_exe(tmp_path_factory):
    import __graft_entry__ as g

    g.build()
    d = tmp_path_factory.mktemp("shim")
    exe = str(d / "shim_smoke")
    libdir = os.path.join(ROOT, "smplpp_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "shim_smoke.cpp"),
           "-o", exe, "-L" + libdir, "-lsmplpp_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    from smplpp_amd import model_io

    path = str(d / "tiny.json")
    model_io.save_model_json(path, model_io.tiny_model(40, seed=3))
    return exe, path


def test_shim_compiles_and_fails_loudly_without_gpu(shim_exe):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    exe, path = shim_exe
    r = subprocess.run([exe, path, "--expect-no-gpu"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and "smplpp::Exception" in r.stdout, r.stdout


@pytest.mark.gpu
def test_shim_runs_fk_and_ik_on_gpu(shim_exe):
    exe, path = shim_exe
    r = subprocess.run([exe, path], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout
