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
    import json

    from smplpp_amd.ik import VPoserDecoder

    vpath = str(d / "vposer.json")
    with open(vpath, "w") as f:
        json.dump({k: v.tolist() for k, v in VPoserDecoder.synthetic_params(seed=3).items()}, f)
    return exe, path, vpath


def test_shim_compiles_and_fails_loudly_without_gpu(shim_exe):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    exe, path, _ = shim_exe
    r = subprocess.run([exe, path, "--expect-no-gpu"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and "smplpp::Exception" in r.stdout, r.stdout


@pytest.mark.gpu
def test_shim_runs_fk_and_ik_on_gpu(shim_exe):
    """FK + IK through the C++ classes, then every method the reference's one caller uses and round 1 lacked
    (IkTask::calcTangents / calcVertexWeights / calcActualPos / calcActualNormal, SMPL::getVertexRaw(index tensor),
    VPoserDecoder::loadParamsFromJson / forward, IkSolver with a VPoser): the numbers the executable prints are compared
    with the Python mirror (smplpp_amd/ik.py, smpl.py) on the same model and inputs."""
    from smplpp_amd import model_io
    from smplpp_amd.ik import IkSolver, VPoserDecoder, convertRotMatToAxisAngle
    from smplpp_amd.smpl import SMPL

    exe, path, vpath = shim_exe
    r = subprocess.run([exe, path, vpath], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout
    got = {ln.split()[0]: np.array(ln.split()[1:], np.float64) for ln in r.stdout.splitlines() if ln and ln.split()[0].isupper()}
    model = model_io.tiny_model(40, seed=3)
    s = SMPL()
    s.setDevice("cuda:0")
    s.init(model)
    theta = np.zeros((1, 25, 3), np.float32)
    flat = theta.reshape(-1)
    for i in range(3, 75):
        flat[i] = np.float32(0.05) * np.sin(np.float32(0.7) * np.float32(i))
    s.launch(np.zeros((1, 10), np.float32), theta)
    fv = model["face_indices"][7].astype(np.int64) - 1
    tri = s.getVertex()[0][fv]
    assert np.abs(got["FACEVERTS"].reshape(3, 3) - tri).max() < 1e-6
    # IkTask-level numbers from the reference's formulas on the Python mirror's vertices / vertex normals
    t1 = tri[1] - tri[0]
    nrm = np.cross(t1, tri[2] - tri[0])
    t2 = np.cross(nrm, t1)
    tang = np.stack([t1 / np.linalg.norm(t1), t2 / np.linalg.norm(t2)], axis=1)  # src/IkTask.cpp:39-45
    assert np.abs(got["TANGENTS"].reshape(3, 2) - tang).max() < 1e-5
    pos = (np.array([0.5, 0.25, 0.25]) @ tri) + tang @ np.array([0.002, -0.001])
    w = np.array([np.linalg.norm(np.cross(tri[(i + 1) % 3] - pos, tri[(i + 2) % 3] - pos)) for i in range(3)])
    w /= w.sum()  # toolbox/GeometryUtils.h:42-52
    assert np.abs(got["WEIGHTS"] - w).max() < 1e-4
    vn = np.stack([np.asarray(s.calcVertexNormal(int(v))).reshape(3) for v in fv])
    an = got["WEIGHTS"] @ vn
    an /= np.linalg.norm(an)  # src/IkTask.cpp:74-86
    assert np.abs(got["ACTUALNORMAL"] - an).max() < 2e-6
    assert np.abs(got["ACTUALPOS"] - (got["WEIGHTS"] @ tri + 0.015 * an)).max() < 2e-6  # :60-72
    # and the engine's own task state for the same weights (on-face point, no normal offset: the weights survive the
    # evaluation's refresh, node.cpp:803-804)
    sol = IkSolver(s, 1, 1)
    onface = np.array([0.2, 0.3, 0.5], np.float32)
    sol.setTasks(face_idx=np.array([7]), vertex_weights=onface[None], target_pos=np.array([[0.1, 0.0, 0.2]], np.float32), phi_limit=np.zeros(1))
    sol.setConfig(np.zeros((1, 10), np.float32), theta)
    sol.eval()
    t = sol.getTasks()
    assert np.abs(t["tangents"][0, 0] - got["TANGENTS"].reshape(3, 2)).max() < 1e-5
    assert np.abs(t["actual_pos"][0, 0] - onface @ tri).max() < 2e-6
    # VPoser
    vp = VPoserDecoder(VPoserDecoder.synthetic_params(seed=3), device=0)
    z = (np.float32(0.1) * np.cos(np.float32(0.37) * np.arange(64, dtype=np.float32))).reshape(2, 32)
    aa, jac = vp.forward(z, want_jac=True)
    assert np.abs(got["VPOSER"] - aa.reshape(-1)).max() < 1e-6
    assert abs(got["VPOSERJACABS"][0] - np.abs(jac.astype(np.float64)).sum()) < 1e-3 * np.abs(jac).sum()
    c, sn = np.cos(np.float32(0.3)), np.sin(np.float32(0.3))
    assert np.abs(got["ROT2AA"] - convertRotMatToAxisAngle(np.array([[c, -sn, 0], [sn, c, 0], [0, 0, 1]], np.float32))[0]).max() < 1e-6
    assert got["LATENTIK"][1] == 44 and np.isfinite(got["LATENTIK"][0]) and np.isfinite(got["LATENTPOS"]).all()
