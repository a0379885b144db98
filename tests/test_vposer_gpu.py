"""GPU parity of the VPoser decoder (src/VPoser.cpp) and of VPoser-latent IK (BASELINE config 5) through the C ABI."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_rotmat_to_axis_angle_sweep_on_gpu():
    """tests/src/TestVPoser.cpp:45-70 through smplpp_rotmat_to_axis_angle, and against the torch restatement."""
    from oracle import vposer_torch as VT
    from smplpp_amd.ik import convertRotMatToAxisAngle
    from test_oracle_vposer import check_against_independent, sweep_matrices

    mats = sweep_matrices()
    aa = convertRotMatToAxisAngle(mats.astype(np.float32))
    assert np.isfinite(aa).all()
    check_against_independent(aa.astype(np.float64), mats)
    ref = VT.convert_rotmat_to_axis_angle(torch.from_numpy(mats.astype(np.float32))).numpy()
    # away from the pi branch the two fp32 evaluations agree closely; near pi sqrt(s + eps) amplifies rounding
    ang = np.linalg.norm(ref, axis=1)
    far = ang < np.pi - 0.05
    assert np.abs(aa[far] - ref[far]).max() < 2e-4
    assert np.abs(np.abs(aa[~far]) - np.abs(ref[~far])).max() < 5e-3


@pytest.fixture(scope="module")
def decoders():
    from oracle import vposer_torch as VT
    from smplpp_amd.ik import VPoserDecoder

    params = VPoserDecoder.synthetic_params()
    return VPoserDecoder(params), VT.VPoserDecoder(params)


def test_decoder_forward_and_jacobian(decoders):
    gpu, ref = decoders
    rng = np.random.default_rng(5)
    z = np.concatenate([rng.random((6, 32)), rng.normal(0, 1.5, (6, 32)), np.zeros((1, 32))]).astype(np.float32)
    out, jac = gpu.forward(z, want_jac=True)
    rout, rjac = ref.forward_with_jacobian(z)
    # fp32 vs fp32: 1.3e-5 rad observed on angles near pi (acos/sqrt conditioning); the IK tolerance is 1e-4 rad
    assert np.abs(out - rout).max() < 5e-5
    assert np.abs(jac - rjac).max() < 2e-4 * max(1.0, np.abs(rjac).max())
    out2 = gpu.forward(z)  # forward only: the activation columns run on the VALU instead of riding in the tangent loops
    assert np.abs(out2 - out).max() < 1e-6


def test_decoder_jacobian_two_frames_per_workgroup(decoders):
    """Batches of more frames than CUs take vposer_jac2_kernel<2>: two frames per workgroup share every weight fragment, the
    layer-0 tangent block is not stored (W1 . diag(s) W0 = 0.99 W1 . (m (.) W0) + 0.01 W1 . W0 with a masked W0 stream and a
    constant product).  601 latents (an odd count: the last workgroup's spare frame must not be stored) against the torch
    restatement on a sample and against the one-frame-per-workgroup instantiation (batches of up to one frame per CU) on all of
    them: the SAME bits (test_decoder_bits_do_not_depend_on_the_shard walks the shard boundaries)."""
    gpu, ref = decoders
    rng = np.random.default_rng(9)
    n = 601
    z = np.concatenate([rng.normal(0, 1.0, (n - 1, 32)), np.zeros((1, 32))]).astype(np.float32)
    out, jac = gpu.forward(z, want_jac=True)
    sel = np.array([0, 1, 2, 299, 300, 598, 599, 600])
    rout, rjac = ref.forward_with_jacobian(z[sel])
    assert np.abs(out[sel] - rout).max() < 5e-5
    assert np.abs(jac[sel] - rjac).max() < 2e-4 * max(1.0, np.abs(rjac).max())
    for lo in range(0, n, 200):
        hi = min(n, lo + 200)
        out1, jac1 = gpu.forward(z[lo:hi], want_jac=True, frame_base=lo)
        assert np.array_equal(out1, out[lo:hi]) and np.array_equal(jac1, jac[lo:hi])


def test_reference_decoder_golden():
    """tests/src/TestVPoser.cpp:72-130 through the C ABI, for anyone holding the license-gated weights (SMPLPP_VPOSER_JSON):
    VPoserDecoder.loadParamsFromJson -> smplpp_vposer_forward.  The value-only kernel is exact fp32 (its summation order is its
    own: 5e-6 on the norm where two torch builds agree to 1e-6); the Jacobian kernel carries fp16x2 operand pieces (2e-5).
    d||out||/dz = jac^T . out / ||out|| against the reference's autograd gradient."""
    from smplpp_amd.ik import VPoserDecoder
    from test_oracle_vposer import reference_decoder_golden

    path, zin, out_gt, grad_gt = reference_decoder_golden()
    vp = VPoserDecoder.loadParamsFromJson(path)
    out = vp.forward(zin)
    assert np.linalg.norm((out - out_gt).ravel()) < 5e-6
    out2, jac = vp.forward(zin, want_jac=True)
    assert np.linalg.norm((out2 - out_gt).ravel()) < 2e-5
    o = out2.reshape(-1, 63).astype(np.float64)
    grad = np.einsum("nr,nrk->nk", o / np.linalg.norm(o, axis=1, keepdims=True), jac.astype(np.float64))
    assert np.linalg.norm((grad - grad_gt).ravel()) < 2e-5 * max(1.0, np.linalg.norm(grad_gt))


def test_latent_ik_eval_and_step(decoders, synth_model, oracle_synth, golden_ik_synth):
    """node.cpp:761-772 + :895-904: theta44 = [pos3 | root3 | z32 | aa22 | aa23]; J over the latent layout is J75 pulled
    back through d(vposer)/dz; the prior adds w_i to A_ii and w_i * theta_i to b_i."""
    from oracle import cpu
    from smplpp_amd.ik import IkSolver
    from smplpp_amd.smpl import SMPL

    gpu, ref = decoders
    g = golden_ik_synth
    K = len(g["face_idx"])
    s = SMPL()
    s.setDevice("cuda:0")
    s.init(synth_model)
    n = 3
    rng = np.random.default_rng(8)
    g44 = np.zeros((n, 44), np.float32)
    g44[:, :3] = [0, 0, 0.05]
    g44[:, 3:6] = rng.normal(0, 0.05, (n, 3))
    g44[:, 6:38] = rng.normal(0, 0.7, (n, 32))
    g44[:, 38:] = rng.normal(0, 0.05, (n, 6))
    sol = IkSolver(s, n, K, vposer=gpu)
    assert sol.theta_dim == 44
    sol.setTasks(face_idx=g["face_idx"], target_pos=g["target_pos"], target_normal=g["target_normal"], phi_limit=np.zeros(K))
    sol.setConfig(np.zeros((n, 10), np.float32), g44)
    e, J = sol.eval()
    assert J.shape == (n, 4 * K, 44 + 2 * K)
    vout, vjac = ref.forward_with_jacobian(g44[:, 6:38])
    for f in range(n):
        th25 = np.zeros((25, 3), np.float32)
        th25[0], th25[1] = g44[f, :3], g44[f, 3:6]
        th25[2:23] = vout[f]
        th25[23], th25[24] = g44[f, 38:41], g44[f, 41:44]
        ts = cpu.TaskSet(g["face_idx"], g["target_pos"], g["target_normal"], phi_limit=np.zeros(K))
        r = oracle_synth.ik_eval(np.zeros(10, np.float32), th25, ts)
        J75 = r["J"]
        Jl = np.concatenate([J75[:, :6], J75[:, 6:69] @ vjac[f].astype(np.float64), J75[:, 69:75], J75[:, 75:]], axis=1)
        de = np.abs(r["e"] - e[f]).reshape(K, 4)
        assert de[:, :3].max() < 5e-6 and de[:, 3].max() < 2e-4  # normal rows: 2 cm triangles (see test_oracle_golden)
        assert np.abs(Jl - J[f]).max() < 1e-3 * max(1.0, np.abs(Jl).max())
        # one step with the prior (node.cpp:895-904)
        A, b = cpu.normal_equations(r["e"], Jl, 44, 2 * K, 0, vposer_theta=g44[f])
        x = cpu.llt_solve(A, b)
        if f == 0:
            x0 = x
    sol.iterate(1)
    _, t44 = sol.getConfig()
    # bars per entry kind (tests/latent_oracle.py): metres / radians on the 12 pass-through entries and on the 63 body angles the
    # decoder emits for the new latent (the north star's 1e-4); the 32 dimensionless latent coordinates (prior weight 1e-5) 1e-4 too
    import latent_oracle as LO

    d_pass, d_ang, d_lat = LO.compare_states(ref, t44[0], g44[0] + x0[:44].astype(np.float32))
    assert d_pass < 1e-4 and d_ang < 1e-4 and d_lat < 1e-4, (d_pass, d_ang, d_lat)
    e2 = sol.iterate(15)
    assert np.isfinite(e2).all()


@pytest.mark.gpu
@pytest.mark.parametrize("K", [20, 25])
def test_latent_jacobian_many_position_only_tasks_in_one_group(decoders, synth_model, oracle_synth, K):
    """The pull-back through d(vposer)/dz runs inside the evaluation's task groups as 16 x 16 tiles on the fp64 matrix pipe, one per
    wavefront (csrc/ik.hip, phase B behind B3).  A group of position-only tasks holds up to 25 of them: 100 rows = 7 row tiles x 2
    column tiles, more than the workgroup has wavefronts — the tile loop's second pass.  Rows against the oracle's J75 pulled back
    through the torch decoder's Jacobian (node.cpp:761-772), entry by entry."""
    from oracle import cpu
    from smplpp_amd.ik import IkSolver, reference_task_faces
    from smplpp_amd.smpl import SMPL

    gpu, ref = decoders
    s = SMPL()
    s.setDevice("cuda:0")
    s.init(synth_model)
    n = 2
    rng = np.random.default_rng(80 + K)
    faces = rng.choice(synth_model["face_indices"].shape[0], K, replace=False).astype(np.int64)
    tp = rng.normal(0, 0.3, (n, K, 3)).astype(np.float32)
    g44 = np.zeros((n, 44), np.float32)
    g44[:, 3:6] = rng.normal(0, 0.05, (n, 3))
    g44[:, 6:38] = rng.normal(0, 0.7, (n, 32))
    g44[:, 38:] = rng.normal(0, 0.05, (n, 6))
    sol = IkSolver(s, n, K, vposer=gpu)
    sol.setTasks(face_idx=faces, target_pos=tp, phi_limit=np.zeros(K), normal_task_weight=np.zeros(K))
    sol.setConfig(np.zeros((n, 10), np.float32), g44)
    e, J = sol.eval()
    assert J.shape == (n, 4 * K, 44 + 2 * K)
    vout, vjac = ref.forward_with_jacobian(g44[:, 6:38])
    import latent_oracle as LO

    for f in range(n):
        ts = cpu.TaskSet(faces, tp[f], phi_limit=np.zeros(K), normal_task_weight=np.zeros(K))
        r = oracle_synth.ik_eval(np.zeros(10, np.float32), LO.splice(g44[f], vout[f]), ts)
        J75 = r["J"]
        Jl = np.concatenate([J75[:, :6], J75[:, 6:69] @ vjac[f].reshape(63, 32).astype(np.float64), J75[:, 69:75], J75[:, 75:]], axis=1)
        assert np.abs(r["e"] - e[f]).max() < 5e-6
        assert np.abs(Jl - J[f]).max() < 1e-3 * max(1.0, np.abs(Jl).max()), (K, f, np.abs(Jl - J[f]).max())
        # the latent block on its own: every (row, latent column), not only the largest entries
        blk = np.abs(Jl[:, 6:38] - J[f][:, 6:38])
        assert blk.max() < 2e-4 * max(1.0, np.abs(Jl[:, 6:38]).max()), (K, f, blk.max())


@pytest.mark.gpu
def test_latent_jacobian_on_a_deep_tree(decoders, synth_model):
    """A kinematic tree of 12 levels runs the evaluation in its second instantiation (EvalPlan<12, 64, 3>), whose ring-vertex region
    is too small for the decoder's Jacobian: there the pull-back takes it from the vertex-normal derivatives' LDS behind B3 instead
    (csrc/ik.hip SVJ_EARLY).  Position + normal rows with offsets, the latent rows against the oracle's J75 pulled back through
    the torch decoder's Jacobian."""
    from oracle import cpu
    from smplpp_amd.ik import IkSolver
    from smplpp_amd.smpl import SMPL
    import latent_oracle as LO

    gpu, ref = decoders
    md = dict(synth_model)
    kt = md["kinematic_tree"].copy()
    kt[0] = np.array([-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 0, 12, 13, 14, 0, 16, 17, 18, 3, 20, 21, 22], np.int64)
    kt[0, 0] = 4294967295
    md["kinematic_tree"] = kt
    s = SMPL()
    s.setDevice("cuda:0")
    s.init(md)
    o = cpu.OracleModel(md)
    rng = np.random.default_rng(93)
    n, K = 2, 7
    faces = rng.integers(0, 13776, (n, K))
    tp = rng.normal(0, 0.4, (n, K, 3)).astype(np.float32)
    tn = rng.normal(0, 1, (n, K, 3)).astype(np.float32)
    tn /= np.linalg.norm(tn, axis=2, keepdims=True)
    bary = rng.dirichlet(np.ones(3), (n, K)).astype(np.float32)
    g44 = np.zeros((n, 44), np.float32)
    g44[:, 3:6] = rng.normal(0, 0.05, (n, 3))
    g44[:, 6:38] = rng.normal(0, 0.5, (n, 32))
    g44[:, 38:] = rng.normal(0, 0.05, (n, 6))
    sol = IkSolver(s, n, K, vposer=gpu)
    sol.setTasks(face_idx=faces, target_pos=tp, target_normal=tn, vertex_weights=bary, phi_limit=np.zeros((n, K)),
                 normal_offset=np.full((n, K), 0.015), normal_task_weight=np.full((n, K), 1.0))
    sol.setConfig(np.zeros((n, 10), np.float32), g44)
    e, J = sol.eval()
    vout, vjac = ref.forward_with_jacobian(g44[:, 6:38])
    for f in range(n):
        ts = cpu.TaskSet(faces[f], tp[f], tn[f], vertex_weights=bary[f], phi_limit=np.zeros(K), normal_offset=np.full(K, 0.015))
        ts.normal_task_weight[:] = 1.0
        r = o.ik_eval(np.zeros(10, np.float32), LO.splice(g44[f], vout[f]), ts)
        J75 = r["J"]
        Jl = np.concatenate([J75[:, :6], J75[:, 6:69] @ vjac[f].reshape(63, 32).astype(np.float64), J75[:, 69:75], J75[:, 75:]], axis=1)
        de = np.abs(r["e"] - e[f]).reshape(K, 4)
        assert de[:, :3].max() < 5e-6 and de[:, 3].max() < 1e-4
        dJ = np.abs(Jl - J[f]).reshape(K, 4, -1)
        scale = max(1.0, np.abs(Jl).max())
        assert dJ[:, :3].max() < 1e-4 * scale and dJ[:, 3].max() < 6e-4 * scale, (f, dJ[:, :3].max(), dJ[:, 3].max())


def test_latent_ik_config4_size_512_frames_50_iterations(decoders, synth_model, oracle_synth):
    """BASELINE.json configs[4] at its stated size on one GPU: 512 frames x 6 position targets x 50 iterations over the 44-d
    VPoser layout (decoder in the loop, prior of node.cpp:895-904).  Frames are re-synchronised with the CPU restatement at
    iterations 1, 10, 25 and 50 — sampled frames plus, at the end, the frames with the LARGEST residual: from the engine's own state (latent
    vector, faces, barycentric weights) one oracle step (oracle FK + analytic J pulled back through the torch decoder's
    Jacobian, fp64 normal equations with the prior, LLT) lands on the engine's next state, same re-projected faces."""
    from oracle import cpu
    from smplpp_amd.ik import IkSolver, reference_task_faces
    from smplpp_amd.smpl import SMPL

    gpu, ref = decoders
    n, K, iters = 512, 6, 50
    s = SMPL()
    s.setDevice("cuda:0")
    s.init(synth_model)
    _, faces = reference_task_faces(K)
    rng = np.random.default_rng(300)
    hid = np.zeros((n, 25, 3), np.float32)
    hid[:, 1:22] = rng.normal(0, 0.15, (n, 21, 3))
    hv = s.launch(np.zeros((n, 10), np.float32), hid, want=("verts",))["verts"]
    tp = hv[:, synth_model["face_indices"][faces] - 1].mean(axis=2)
    sol = IkSolver(s, n, K, vposer=gpu)
    sol.setTasks(face_idx=faces, target_pos=tp, phi_limit=np.zeros(K), normal_task_weight=np.zeros(K))
    sol.setConfig(np.zeros((n, 10), np.float32), np.zeros((n, 44), np.float32))

    def oracle_step(f, g44, t_before):
        vout, vjac = ref.forward_with_jacobian(g44[None, 6:38])
        th25 = np.zeros((25, 3), np.float32)
        th25[0], th25[1] = g44[:3], g44[3:6]
        th25[2:23] = vout[0]
        th25[23], th25[24] = g44[38:41], g44[41:44]
        ts = cpu.TaskSet(t_before["face_idx"][f], tp[f], phi_limit=np.zeros(K), normal_task_weight=np.zeros(K),
                         vertex_weights=t_before["vertex_weights"][f])
        r = oracle_synth.ik_eval(np.zeros(10, np.float32), th25, ts)
        J75 = r["J"]
        Jl = np.concatenate([J75[:, :6], J75[:, 6:69] @ vjac[0].astype(np.float64), J75[:, 69:75], J75[:, 75:]], axis=1)
        A, b = cpu.normal_equations(r["e"], Jl, 44, 2 * K, 0, vposer_theta=g44)
        return cpu.llt_solve(A, b)[:44], float(r["e"] @ r["e"])

    import latent_oracle as LO

    sample = [0, 91, 300, 511]
    done = 0
    worst = np.zeros(3)
    for target in (1, 10, 25, 50):
        if target - 1 > done:
            sol.iterate(target - 1 - done)
            done = target - 1
        _, g_before = sol.getConfig()
        t_before = sol.getTasks()
        e2 = sol.iterate(1)
        done += 1
        if target == 1:
            e2_first = e2.copy()
        _, g_after = sol.getConfig()
        check = list(sample)
        if target == iters:
            check += [int(f) for f in np.argsort(-e2)[:3]]
        for f in check:
            x, e2o = oracle_step(f, g_before[f], t_before)
            # per entry kind (tests/latent_oracle.py:compare_states): the 12 pass-through entries (metres / radians) and the 63
            # body angles decoded from the new latent within the north star's 1e-4; the 32 latent coordinates (dimensionless,
            # held by a 1e-5 prior only) within 1e-4 as well (measured 3.6e-7)
            d_pass, d_ang, d_lat = LO.compare_states(ref, g_after[f], g_before[f] + x.astype(np.float32))
            worst = np.maximum(worst, (d_pass, d_ang, d_lat))
            assert d_pass < 1e-4 and d_ang < 1e-4 and d_lat < 1e-4, (target, f, d_pass, d_ang, d_lat)
            assert abs(e2o - e2[f]) < 2e-5 * max(1.0, e2o), (target, f)
    assert done == iters
    print("configs[4] steps: worst |d| pass-through %.3g, decoded angles %.3g rad, latent %.3g" % tuple(worst))
    assert np.isfinite(e2).all() and np.isfinite(g_after).all()
    # (with random decoder weights the targets — poses drawn in joint-angle space — lie outside the decoder's range: the solves
    # settle at residuals of 3e-3..6e-2 instead of converging, which is what makes their last steps worth checking above)
    assert (e2 < e2_first).mean() > 0.9


def test_decoder_bits_do_not_depend_on_the_shard(decoders):
    """a latent must decode — value AND Jacobian — to the same bits whether the job runs as one 512-frame
    batch or cut into shards (BASELINE configs[4]: 512 frames over 4 GPUs).  The Jacobian kernel rotates its k loop by the frame's
    GLOBAL group (frame_base + local index) / 2, and the one-frame-per-workgroup instantiation (shards of fewer frames than CUs)
    does the same arithmetic per frame as the two-frame one.  Shards starting on even AND odd global indices, and a one-frame shard."""
    gpu, _ = decoders
    rng = np.random.default_rng(31)
    n = 512
    z = rng.normal(0, 1.0, (n, 32)).astype(np.float32)
    out, jac = gpu.forward(z, want_jac=True)
    for lo, hi in [(128, 256), (0, 128), (129, 256), (255, 512), (300, 301), (1, 512)]:
        o, j = gpu.forward(z[lo:hi], want_jac=True, frame_base=lo)
        assert np.array_equal(o, out[lo:hi]), (lo, hi)
        assert np.array_equal(j, jac[lo:hi]), (lo, hi)
    # the guard that makes the test meaningful: with a WRONG base the k loop starts elsewhere (5 * group mod 32: a base of 2 moves
    # every group by one; 0 would not do — 128 / 2 groups is a multiple of 32) and the last bits differ somewhere, nothing more
    o0, j0 = gpu.forward(z[128:256], want_jac=True, frame_base=2)
    assert np.abs(j0 - jac[128:256]).max() < 1e-4
    assert not np.array_equal(j0, jac[128:256])


def test_latent_ik_trajectory_does_not_depend_on_the_shard(decoders, synth_model):
    """The same property through the whole latent IK loop (decoder -> FK -> evaluation -> solve -> re-projection), 8 iterations:
    frames 256..383 of a 512-frame job == the same frames as a 128-frame shard with frame_base = 256, bit for bit."""
    from smplpp_amd.ik import IkSolver, reference_task_faces
    from smplpp_amd.smpl import SMPL

    gpu, _ = decoders
    s = SMPL()
    s.setDevice("cuda:0")
    s.init(synth_model)
    K, n = 6, 512
    _, faces = reference_task_faces(K)
    rng = np.random.default_rng(33)
    hid = np.zeros((n, 25, 3), np.float32)
    hid[:, 1:22] = rng.normal(0, 0.15, (n, 21, 3))
    hv = s.launch(np.zeros((n, 10), np.float32), hid, want=("verts",))["verts"]
    tp = hv[:, synth_model["face_indices"][faces] - 1].mean(axis=2)
    g0 = np.zeros((n, 44), np.float32)
    g0[:, 6:38] = rng.normal(0, 0.3, (n, 32))

    def run(lo, hi):
        sol = IkSolver(s, hi - lo, K, vposer=gpu, frame_base=lo)
        sol.setTasks(face_idx=faces, target_pos=np.ascontiguousarray(tp[lo:hi]), phi_limit=np.zeros(K), normal_task_weight=np.zeros(K))
        sol.setConfig(np.zeros((hi - lo, 10), np.float32), np.ascontiguousarray(g0[lo:hi]))
        e2 = sol.iterate(8)
        _, th = sol.getConfig()
        t = sol.getTasks()
        return e2, th, t["face_idx"], t["vertex_weights"]

    full = run(0, n)
    for lo, hi in [(256, 384), (383, 512)]:
        part = run(lo, hi)
        for a, b in zip(part, full):
            assert np.array_equal(a, b[lo:hi]), (lo, hi)


def test_value_only_instantiation_decodes_the_jacobian_kernels_bits(decoders):
    """vposer_jac2_kernel<NF, true> (round 5): the value path of the Jacobian kernel alone, used by the capture loops to have theta25
    early while the Jacobian is made beside the pose step and the fused kernel.  The decoded angles must be the SAME BITS a call with
    the Jacobian writes — in both instantiations (one and two frames per workgroup) and in shards."""
    import ctypes as C

    from smplpp_amd import _lib

    gpu, _ = decoders
    L = _lib.load()
    L.smplpp_debug_vposer_value.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    L.smplpp_debug_vposer_value.restype = C.c_int
    rng = np.random.default_rng(41)
    for n, base in ((5, 0), (64, 0), (64, 7), (300, 0), (513, 128)):
        z = rng.normal(0, 1.0, (n, 32)).astype(np.float32)
        out, _jac = gpu.forward(z, want_jac=True, frame_base=base)
        val = np.full((n, 21, 3), np.nan, np.float32)
        assert L.smplpp_debug_vposer_value(gpu._h, n, base, z.ctypes.data_as(C.c_void_p), val.ctypes.data_as(C.c_void_p)) == 0, L.smplpp_last_error()
        assert np.array_equal(val, out), (n, base, float(np.abs(val - out).max()))


def test_decoder_jacobian_is_finite_at_the_axis_angle_branch_points():
    """tests/src/TestVPoser.cpp:36-43 on the GPU path: the reference checks that the gradient of convertRotMatToAxisAngle has no NaN
    at the identity (angle 0: the Taylor branch, src/VPoser.cpp:105-111) and at a rotation by pi (the sqrt-diagonal branch, :53-103,
    kept finite by the (1 - eps) shrink inside acos and the eps under the sqrt, :41, :60).  Here the decoder's last layer is
    biased so that, at z = 0, its 21 joints decode EXACTLY to those rotations — identity, pi about x / y / z and about oblique axes,
    and rotations 1e-4 rad and 1e-3 rad from either branch point — with small random weights in front, so that d(out)/dz is the
    derivative of the axis-angle conversion times a non-zero matrix: values and Jacobian must be finite, the decoded rotations
    must be the intended ones, and both must agree with the torch restatement (whose backward() is the reference's own formula)."""
    from scipy.spatial.transform import Rotation
    from oracle import vposer_torch as VT
    from smplpp_amd.ik import VPoserDecoder

    rng = np.random.default_rng(12)
    axes = [np.array(a, np.float64) / np.linalg.norm(a) for a in
            ([1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [1, -2, 3], [0.3, 0.5, -0.8], [-1, 0.2, 0.1])]
    rots = [np.eye(3)]
    rots += [Rotation.from_rotvec(np.pi * a).as_matrix() for a in axes]                      # exactly pi
    rots += [Rotation.from_rotvec((np.pi - d) * a).as_matrix() for a in axes[:3] for d in (1e-4, 1e-3)]  # just below pi
    rots += [Rotation.from_rotvec(d * a).as_matrix() for a in axes[3:6] for d in (1e-4, 1e-3)]          # just above 0
    rots += [Rotation.from_rotvec(0.7 * axes[4]).as_matrix()]                                # a generic one
    assert len(rots) == 21
    R = np.stack(rots)
    params = VPoserDecoder.synthetic_params(seed=8)
    params["decoder_net.5.weight"] = (params["decoder_net.5.weight"] * np.float32(0.05)).astype(np.float32)
    # 6D representation: the rotation's first two columns, stored [3, 2] per joint (src/VPoser.cpp:129-141); the bias is the whole
    # output at z = 0 once the hidden layers' contribution is taken off: solve for it with the torch restatement
    target = R[:, :, :2].reshape(-1).astype(np.float32)
    params["decoder_net.5.bias"] = np.zeros(126, np.float32)
    hidden0 = VT.VPoserDecoder(params).net(torch.zeros(1, 32)).detach().numpy()[0]
    params["decoder_net.5.bias"] = (target - hidden0).astype(np.float32)
    gpu, ref = VPoserDecoder(params), VT.VPoserDecoder(params)
    z = np.zeros((3, 32), np.float32)
    z[1] = rng.normal(0, 1e-3, 32)  # a hair off the branch points
    z[2] = rng.normal(0, 0.3, 32)
    out, jac = gpu.forward(z, want_jac=True)
    assert np.isfinite(out).all() and np.isfinite(jac).all()
    # Exactly pi about an axis with components of BOTH signs (joints 5, 6, 7) is where the reference's own sign fixes
    # (src/VPoser.cpp:62-103) return the axis-angle of a DIFFERENT rotation (the torch restatement, run on the CPU: 0.4-1.0 off the
    # intended matrix) and where one fp32 ulp decides which fix fires: there only finiteness is asserted.  Everywhere else the
    # decoded rotation is the intended one (7e-4 at pi: acos((1 - eps) x) and sqrt(s + eps) bound the angle away from pi) and the
    # engine agrees with the restatement.
    ok = np.array([j for j in range(21) if j not in (5, 6, 7)])
    got = Rotation.from_rotvec(out[0].astype(np.float64)).as_matrix()
    assert np.abs(got[ok] - R[ok]).max() < 2e-3
    rout, rjac = ref.forward_with_jacobian(z)
    assert np.isfinite(rjac).all()
    same = np.abs(Rotation.from_rotvec(out[:, ok].reshape(-1, 3).astype(np.float64)).as_matrix()
                  - Rotation.from_rotvec(rout[:, ok].reshape(-1, 3).astype(np.float64)).as_matrix()).max()
    assert same < 2e-3
    near0 = [0] + list(range(14, 21))
    assert np.abs(out[:, near0] - rout[:, near0]).max() < 5e-6
    rows = np.concatenate([np.arange(3 * j, 3 * j + 3) for j in near0])
    assert np.abs(jac[:, rows] - rjac[:, rows]).max() < 2e-4 * max(1.0, np.abs(rjac[:, rows]).max())
