"""The header-only C++ shim (include/smplpp/SMPL.h, IkTask.h, VPoser.h, Tensor.h: the reference's class names and signatures over
the C ABI) compiles with plain g++ and links libsmplpp_hip.so; without a GPU it fails loudly with smplpp::Exception.  With one:
tests/cpp/shim_smoke.cpp runs FK, the IkTask methods and the VPoser decoder, and tests/cpp/node_loop.cpp runs the reference's
loop body (node/node.cpp:645-1001) and frame advance (:1369-1407) transcribed onto the shim.  Everything the executables print or
write is checked against the CPU ORACLE (oracle/), the reference-generated golden trajectory (tests/golden/ik_traj50.npz) and the
torch restatement of the decoder — never against the Python mirror of the same library."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compile(src, exe):
    libdir = os.path.join(ROOT, "smplpp_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", src),
           "-o", exe, "-L" + libdir, "-lsmplpp_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)


@pytest.fixture(scope="module")
def shim_exe(tmp_path_factory):
    import __graft_entry__ as g

    g.build()
    d = tmp_path_factory.mktemp("shim")
    exe = str(d / "shim_smoke")
    _compile("shim_smoke.cpp", exe)
    from smplpp_amd import model_io

    path = str(d / "tiny.json")
    model_io.save_model_json(path, model_io.tiny_model(40, seed=3))
    import json

    from smplpp_amd.ik import VPoserDecoder

    vpath = str(d / "vposer.json")
    with open(vpath, "w") as f:
        json.dump({k: v.tolist() for k, v in VPoserDecoder.synthetic_params(seed=3).items()}, f)
    return exe, path, vpath


@pytest.fixture(scope="module")
def node_exe(tmp_path_factory, synth_model):
    import __graft_entry__ as g

    g.build()
    d = tmp_path_factory.mktemp("node")
    exe = str(d / "node_loop")
    _compile("node_loop.cpp", exe)
    from smplpp_amd import model_io

    path = str(d / "model.json")  # the full-size stand-in in the reference's .json schema (scripts/preprocess.py:98-117)
    model_io.save_model_json(path, synth_model)
    return exe, path, d


def test_shim_compiles_and_fails_loudly_without_gpu(shim_exe):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    exe, path, _ = shim_exe
    r = subprocess.run([exe, path, "--expect-no-gpu"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and "smplpp::Exception" in r.stdout, r.stdout


def test_node_loop_compiles_against_the_shim(tmp_path):
    """The transcription of node/node.cpp's loop (tests/cpp/node_loop.cpp) compiles against the shim headers with -Wall -Werror and
    links the C ABI — the compile-time half of INTEGRATION.md's edit table (no GPU needed)."""
    import __graft_entry__ as g

    g.build()
    _compile("node_loop.cpp", str(tmp_path / "node_loop"))


@pytest.mark.gpu
def test_shim_runs_fk_and_ik_on_gpu(shim_exe):
    """FK + IK through the C++ classes, then every method the reference's one caller uses (IkTask::calcTangents /
    calcVertexWeights / calcActualPos / calcActualNormal, SMPL::getVertexRaw(index tensor), VPoserDecoder::loadParamsFromJson /
    forward, IkSolver with a VPoser).  The numbers the executable prints are compared with the CPU oracle's vertices and vertex
    normals (oracle/smpl_oracle.c) pushed through the reference's formulas in numpy, and with the torch restatement of the
    decoder (oracle/vposer_torch.py)."""
    from oracle import cpu
    from oracle.vposer_torch import VPoserDecoder as TorchDecoder, convert_rotmat_to_axis_angle
    from smplpp_amd import model_io
    from smplpp_amd.ik import VPoserDecoder

    exe, path, vpath = shim_exe
    r = subprocess.run([exe, path, vpath], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout
    got = {ln.split()[0]: np.array(ln.split()[1:], np.float64) for ln in r.stdout.splitlines() if ln and ln.split()[0].isupper()}
    model = model_io.tiny_model(40, seed=3)
    O = cpu.OracleModel(model)
    theta = np.zeros((1, 25, 3), np.float32)
    flat = theta.reshape(-1)
    for i in range(3, 75):
        flat[i] = np.float32(0.05) * np.sin(np.float32(0.7) * np.float32(i))
    verts = O.fk(np.zeros((1, 10), np.float32), theta, want=("verts",))["verts"][0]
    fv = model["face_indices"][7].astype(np.int64) - 1
    tri = verts[fv].astype(np.float64)
    assert np.abs(got["FACEVERTS"].reshape(3, 3) - tri).max() < 1e-6
    # IkTask-level numbers: the reference's formulas on the ORACLE's vertices / vertex normals
    t1 = tri[1] - tri[0]
    nrm = np.cross(t1, tri[2] - tri[0])
    t2 = np.cross(nrm, t1)
    tang = np.stack([t1 / np.linalg.norm(t1), t2 / np.linalg.norm(t2)], axis=1)  # src/IkTask.cpp:39-45
    assert np.abs(got["TANGENTS"].reshape(3, 2) - tang).max() < 1e-5
    pos = (np.array([0.5, 0.25, 0.25]) @ tri) + tang @ np.array([0.002, -0.001])
    w = np.array([np.linalg.norm(np.cross(tri[(i + 1) % 3] - pos, tri[(i + 2) % 3] - pos)) for i in range(3)])
    w /= w.sum()  # toolbox/GeometryUtils.h:42-52
    assert np.abs(got["WEIGHTS"] - w).max() < 1e-4
    vn = np.stack([O.vertex_normal(verts, int(v)) for v in fv]).astype(np.float64)
    an = got["WEIGHTS"] @ vn
    an /= np.linalg.norm(an)  # src/IkTask.cpp:74-86
    assert np.abs(got["ACTUALNORMAL"] - an).max() < 2e-6
    assert np.abs(got["ACTUALPOS"] - (got["WEIGHTS"] @ tri + 0.015 * an)).max() < 2e-6  # :60-72
    # VPoser: the torch restatement of src/VPoser.cpp on the same weights
    import torch

    params = VPoserDecoder.synthetic_params(seed=3)
    dec = TorchDecoder(params)
    z = (np.float32(0.1) * np.cos(np.float32(0.37) * np.arange(64, dtype=np.float32))).reshape(2, 32)
    aa, jac = dec.forward_with_jacobian(z)
    assert np.abs(got["VPOSER"] - np.asarray(aa).reshape(-1)).max() < 2e-6
    assert abs(got["VPOSERJACABS"][0] - np.abs(np.asarray(jac, np.float64)).sum()) < 1e-3 * np.abs(np.asarray(jac)).sum()
    c, sn = np.cos(np.float32(0.3)), np.sin(np.float32(0.3))
    rot = torch.tensor([[[c, -sn, 0], [sn, c, 0], [0, 0, 1]]], dtype=torch.float32)
    assert np.abs(got["ROT2AA"] - convert_rotmat_to_axis_angle(rot).numpy()[0]).max() < 1e-6
    assert got["LATENTIK"][1] == 44 and np.isfinite(got["LATENTIK"][0]) and np.isfinite(got["LATENTPOS"]).all()


def _task_header(names, faces, tpos, tnrm):
    b = struct.pack("<q", len(names))
    for nm, f, p, n in zip(names, faces, tpos, tnrm):
        e = nm.encode()
        b += struct.pack("<q", len(e)) + e + struct.pack("<q", int(f)) + np.asarray(p, "<f4").tobytes() + np.asarray(n, "<f4").tobytes()
    return b


def _read_states(buf, off, count, K, extra_i64=0):
    """`count` records of [extra int64s][theta 75 f32][faces K i64][weights 3K f32][e2 f64] -> arrays"""
    th, fc, wt, e2, ex = [], [], [], [], []
    for _ in range(count):
        if extra_i64:
            ex.append(np.frombuffer(buf, "<i8", extra_i64, off).copy()); off += 8 * extra_i64
        th.append(np.frombuffer(buf, "<f4", 75, off).reshape(25, 3).copy()); off += 300
        fc.append(np.frombuffer(buf, "<i8", K, off).copy()); off += 8 * K
        wt.append(np.frombuffer(buf, "<f4", 3 * K, off).reshape(K, 3).copy()); off += 12 * K
        e2.append(struct.unpack_from("<d", buf, off)[0]); off += 8
    return np.array(th), np.array(fc), np.array(wt), np.array(e2), ex, off


@pytest.mark.gpu
def test_node_loop_reproduces_the_reference_trajectory(node_exe, synth_model, oracle_synth):
    """The loop body of node/node.cpp:645-1001 transcribed onto the shim (tests/cpp/node_loop.cpp), fed the 50 states of the
    reference-generated trajectory (tests/golden/ik_traj50.npz): one pass from each state lands within 1e-4 rad of the reference's
    next state, on the reference's faces; free-running from state 0 it stays within the reference's own thread-count divergence
    for the first steps.  Then the getters of the node's other call sites (:121-123, 183-186, 214-220, 976-978), IkTask's methods
    and SMPL::out through a copy of the model object — against the oracle on the pose the loop ended in."""
    exe, model_json, d = node_exe
    g = np.load(os.path.join(GOLDEN, "ik_traj50.npz"))
    K = len(g["face_idx"])
    names = ["task%02d" % k for k in range(K)]  # std::map order = the golden's task order
    S = 50
    inp = struct.pack("<q", 0) + _task_header(names, g["face_idx"], g["target_pos"], g["target_normal"]) + struct.pack("<q", S)
    for s in range(S):
        inp += g["traj_theta"][s].astype("<f4").tobytes() + g["traj_faces"][s].astype("<i8").tobytes() + g["traj_weights"][s].astype("<f4").tobytes()
    fin, fout = str(d / "traj_in.bin"), str(d / "traj_out.bin")
    open(fin, "wb").write(inp)
    r = subprocess.run([exe, model_json, fin, fout], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout
    buf = open(fout, "rb").read()
    assert struct.unpack_from("<q", buf, 0)[0] == S
    th, fc, wt, e2, _, off = _read_states(buf, 8, S, K)
    for it in range(S):
        assert np.abs(th[it] - g["traj_theta"][it + 1]).max() < 1e-4, it
        assert (fc[it] == g["traj_faces"][it + 1]).all(), it
        assert abs(e2[it] - g["traj_e_sqnorm"][it]) < 2e-5 * max(1.0, e2[it]), it
    thf, fcf, wtf, e2f, _, off = _read_states(buf, off, S, K)
    ref_noise = np.abs(g["alt_theta"] - g["traj_theta"]).reshape(51, -1).max(axis=1)
    d_ref = np.abs(thf - g["traj_theta"][1:]).reshape(S, -1).max(axis=1)
    assert d_ref[:5].max() < 1e-4 and d_ref.max() < 10.0 * ref_noise.max() and e2f[-1] < 1e-3
    # ---- getters / IkTask methods / SMPL::out on the last pose, against the oracle
    V, F, face0 = struct.unpack_from("<3q", buf, off); off += 24
    fvs = np.frombuffer(buf, "<f8", 9, off).reshape(3, 3); off += 72
    fidx = np.frombuffer(buf, "<i8", 3, off); off += 24
    vidx, adj_n, adj_sum = struct.unpack_from("<3q", buf, off); off += 24
    tang = np.frombuffer(buf, "<f4", 6, off).reshape(3, 2); off += 24
    apos = np.frombuffer(buf, "<f4", 3, off); off += 12
    anrm = np.frombuffer(buf, "<f4", 3, off); off += 12
    nerr = struct.unpack_from("<d", buf, off)[0]; off += 8
    assert off == len(buf)
    assert V == 6890 and F == 13776 and face0 == fcf[-1][0]
    last = thf[-2]  # the pose of the LAST launch: the final pass launched with the state the pass before it produced (node.cpp:777)
    verts = oracle_synth.fk(np.zeros((1, 10), np.float32), last[None], want=("verts",))["verts"][0]
    tri_ids = synth_model["face_indices"][face0].astype(np.int64) - 1
    assert (fidx == tri_ids).all() and vidx == tri_ids[0]
    assert np.abs(fvs - verts[tri_ids]).max() < 1e-5
    faces_of_v = np.where((synth_model["face_indices"] - 1 == vidx).any(axis=1))[0]  # src/SMPL.cpp:620-640
    assert adj_n == len(faces_of_v) and adj_sum == int(faces_of_v.sum())
    tri = verts[tri_ids].astype(np.float64)
    t1 = tri[1] - tri[0]
    t2 = np.cross(np.cross(t1, tri[2] - tri[0]), t1)
    assert np.abs(tang - np.stack([t1 / np.linalg.norm(t1), t2 / np.linalg.norm(t2)], axis=1)).max() < 2e-5
    w = wtf[-1][0].astype(np.float64)  # the first task's weights after the free run
    vn = np.stack([oracle_synth.vertex_normal(verts, int(v)) for v in tri_ids]).astype(np.float64)
    an = w @ vn
    an /= np.linalg.norm(an)
    assert np.abs(anrm - an).max() < 5e-6 and np.abs(apos - w @ tri).max() < 5e-6  # normalOffset_ 0: src/IkTask.cpp:60-72
    assert abs(nerr - (float(an @ g["target_normal"][0]) + 1.0)) < 1e-5  # node/node.cpp:811-812
    obj = open(fout + ".obj").read().splitlines()  # SMPL::out, src/SMPL.cpp:757-790
    assert len(obj) == 6890 + 13776 and obj[0].startswith("v ") and obj[6890].startswith("f ")
    v0 = np.array(obj[0].split()[1:], np.float64)
    assert np.abs(v0 - verts[0]).max() < 1e-5 and obj[6890].split()[1:] == [str(int(x)) for x in synth_model["face_indices"][0]]


@pytest.mark.gpu
def test_node_loop_capture_window_step_by_step_vs_oracle(node_exe, oracle_synth):
    """solveMocapMotion on the shim: 41 Baseline markers in std::map (name) order, the per-iteration target switch of
    node/node.cpp:667-700 (missing marker: weight 0, target 0), the skip rule of :785, 32 iterations on the first frame and then one
    per frame (:1369-1407), over frames 400..599 of sample_walk.c3d (a window with missing markers and a frame without any).  The
    program records the state in front of and behind every pass; the oracle repeats single passes from the recorded states:
    1e-4 rad.  The stored motion (motionMsg.data_list) must be one instant per frame, in order."""
    from oracle import cpu
    from smplpp_amd import mocap

    exe, model_json, d = node_exe
    g = np.load(os.path.join(GOLDEN, "sample_walk_full.npz"))
    order = np.argsort(np.array([str(n) for n in g["task_names"]]))  # std::map<std::string, IkTask>: lexicographic
    names = [str(g["task_names"][i]) for i in order]
    assert names == sorted(names)
    faces = np.array([mocap.BASELINE41[nm] for nm in names], np.int64)
    w0, T = 400, 200
    pts = (g["points"][w0:w0 + T] - g["points"][0][g["valid"][0]].mean(axis=0) + np.array([0, -0.3, 0], np.float32)).astype(np.float32)[:, order]
    valid = g["valid"][w0:w0 + T][:, order]
    K = len(names)
    nv = valid.sum(axis=1)
    assert (nv < K).any() and (nv < K // 2).any()
    theta0 = np.zeros((25, 3), np.float32)
    warm = 32
    inp = (struct.pack("<q", 1) + _task_header(names, faces, np.zeros((K, 3)), np.tile([0, 0, 1.0], (K, 1))) + struct.pack("<q", T)
           + pts.astype("<f4").tobytes() + valid.astype("<i8").tobytes() + struct.pack("<q", warm) + theta0.astype("<f4").tobytes())
    fin, fout = str(d / "cap_in.bin"), str(d / "cap_out.bin")
    open(fin, "wb").write(inp)
    r = subprocess.run([exe, model_json, fin, fout], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout
    buf = open(fout, "rb").read()
    off, passes = 0, []
    while True:
        frame = struct.unpack_from("<q", buf, off)[0]; off += 8
        if frame < 0:
            break
        rec = []
        for _ in range(2):
            th = np.frombuffer(buf, "<f4", 75, off).reshape(25, 3).copy(); off += 300
            fc = np.frombuffer(buf, "<i8", K, off).copy(); off += 8 * K
            wt = np.frombuffer(buf, "<f4", 3 * K, off).reshape(K, 3).copy(); off += 12 * K
            rec.append((th, fc, wt))
        e2 = struct.unpack_from("<d", buf, off)[0]; off += 8
        passes.append((frame, rec[0], rec[1], e2))
    iters, stored = struct.unpack_from("<2q", buf, off); off += 16
    motion = np.frombuffer(buf, "<f4", stored * 76, off).reshape(stored, 76)
    assert iters == len(passes) == warm - 1 + T and stored == T
    assert (motion[:, 0] == np.arange(T)).all()  # one instant per frame, in order (node.cpp:1389-1398)
    assert [p[0] for p in passes] == [0] * (warm - 1) + list(range(T))
    gap = [t for t in range(T) if nv[t] < K // 2]
    missing = [t for t in range(1, T) if K // 2 <= nv[t] < K]
    check_frames = sorted(set([1, 2, 50, 120, 199] + gap + [t + 1 for t in gap if t + 1 < T] + missing[:4] + missing[-2:]))
    beta = np.zeros(10, np.float32)
    worst, checked = 0.0, 0
    for idx, (frame, before, after, e2) in enumerate(passes):
        if idx >= 3 and not (idx >= warm - 1 and frame in check_frames):
            continue  # (the first three passes of the warm-up, then the listed frames)
        v = valid[frame]
        if nv[frame] < K // 2:  # node.cpp:785: the solve block is skipped, the configuration stands
            assert e2 == -1.0 and np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1]), frame
            continue
        tp = np.where(v[:, None], pts[frame], 0.0).astype(np.float32)
        ts = cpu.TaskSet(before[1], tp, phi_limit=np.zeros(K), normal_offset=np.full(K, 0.015), vertex_weights=before[2])
        ts.pos_task_weight[:] = v.astype(np.float64)
        ts.normal_task_weight[:] = 0.0
        _, tho, _ = oracle_synth.ik_solve(beta, before[0], ts, 1, enable_qp=True)
        dd = float(np.abs(tho - after[0]).max())
        worst = max(worst, dd)
        checked += 1
        assert dd < 1e-4, (idx, frame, int(nv[frame]), dd)
        assert np.array_equal(motion[frame, 1:].reshape(25, 3), after[0]) if idx >= warm - 1 else True
    assert checked >= 10 and worst > 0.0
