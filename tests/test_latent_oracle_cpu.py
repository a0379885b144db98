"""tests/latent_oracle.py is what the GPU parity tests of the latent layout compare the engine with: here it is checked ON ITS OWN,
on the CPU — its Jacobian (the oracle's direct Jacobian pulled back through the torch decoder's autograd Jacobian, node.cpp:761-772)
against central differences of the oracle's residual in the 44-d vector, and the a17 equivalence it relies on (phi pinned: the box
QP's optimum is the LLT solution)."""
import numpy as np
import pytest

import latent_oracle as LO


@pytest.fixture(scope="module")
def setup(oracle_synth):
    from oracle import cpu, vposer_torch as VT
    from smplpp_amd import mocap
    from smplpp_amd.ik import VPoserDecoder

    ref = VT.VPoserDecoder(VPoserDecoder.synthetic_params())
    names = sorted(mocap.BASELINE41)[:9]
    faces = np.array([mocap.BASELINE41[n] for n in names], np.int64)
    K = len(faces)
    rng = np.random.default_rng(3)
    g = np.zeros(44, np.float32)
    g[:3] = [0.05, -0.1, 0.4]
    g[3:6] = rng.normal(0, 0.2, 3)
    g[6:38] = rng.normal(0, 0.8, 32)
    g[38:] = rng.normal(0, 0.2, 6)
    tp = rng.normal(0, 0.4, (K, 3)).astype(np.float32)
    ts = cpu.TaskSet(faces, tp, phi_limit=np.zeros(K), normal_offset=np.full(K, 0.015), normal_task_weight=np.zeros(K))
    return oracle_synth, ref, g, ts


def _residual(oracle, ref, g44, ts):
    import torch

    with torch.no_grad():
        vout = ref.forward(torch.from_numpy(np.ascontiguousarray(g44[None, LO.LATENT]))).numpy()[0]
    return oracle.ik_eval(np.zeros(10, np.float32), LO.splice(g44, vout), ts.copy())["e"]


def test_latent_jacobian_against_central_differences(setup):
    oracle, ref, g, ts = setup
    K = ts.K
    vout, vjac = ref.forward_with_jacobian(g[None, LO.LATENT])
    r = oracle.ik_eval(np.zeros(10, np.float32), LO.splice(g, vout[0]), ts.copy())
    J75 = r["J"]
    Jl = np.concatenate([J75[:, :6], J75[:, 6:69] @ vjac[0].reshape(63, 32).astype(np.float64), J75[:, 69:75], J75[:, 75:]], axis=1)
    assert Jl.shape == (4 * K, 44 + 2 * K)
    scale = max(1.0, np.abs(Jl[:, :44]).max())
    for i in (0, 2, 4, 5, 6, 13, 20, 37, 38, 43):  # translation, root rotation, latent coordinates, the pass-through joints
        h = 2e-3
        gp, gm = g.copy(), g.copy()
        gp[i] += h
        gm[i] -= h
        fd = (_residual(oracle, ref, gp, ts) - _residual(oracle, ref, gm, ts)) / (float(gp[i]) - float(gm[i]))
        # (the residual is fp32 FK through a decoder: central differences at h = 2e-3 carry ~1e-3 of truncation + rounding)
        assert np.abs(fd - Jl[:, i]).max() < 5e-3 * scale, (i, float(np.abs(fd - Jl[:, i]).max()))
    # the latent columns are not trivially zero: the decoder's gain on this latent shows in the rows
    assert np.abs(Jl[:, 6:38]).max() > 1e-3


def test_box_qp_equals_llt_when_phi_is_pinned(setup):
    """phiLimit_ = 0 zeroes the phi columns of J, A is block-diagonal in phi, the QP optimum is the LLT solution."""
    oracle, ref, g, ts = setup
    a = LO.latent_step(oracle, ref, np.zeros(10), g, ts, enable_qp=True, project=False)
    b = LO.latent_step(oracle, ref, np.zeros(10), g, ts, enable_qp=False, project=False)
    assert np.abs(a["x"] - b["x"]).max() < 1e-12 and np.abs(a["x"][44:]).max() == 0.0
    assert LO.compare_states(ref, a["g44"], b["g44"]) == (0.0, 0.0, 0.0)
    # with phi and beta live the boxes hold
    ts2 = ts.copy()
    ts2.phi_limit[:] = 0.04
    c = LO.latent_step(oracle, ref, np.zeros(10), g, ts2, enable_qp=True, optimize_beta=True, project=True)
    K = ts.K
    assert np.abs(c["x"][44: 44 + 2 * K]).max() <= 0.04 + 1e-12 and np.abs(c["x"][44 + 2 * K:]).max() <= 0.5 + 1e-12
    assert c["face_idx"].shape == (K,) and np.isfinite(c["closest"]).all()
