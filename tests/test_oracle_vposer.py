"""The VPoser restatement (oracle/vposer_torch.py) against the reference's self-contained sweep
(tests/src/TestVPoser.cpp:16-70): identity; angles {0, pi/4, pi/2, pi} +- {0, 1e-12 ... 1e-1} about each axis; 10 000
random unit quaternions; ||aa - aa_ref|| < 5e-3 (sign flip allowed at pi); output and gradient NaN-free."""
import numpy as np
import torch
from scipy.spatial.transform import Rotation

from oracle import vposer_torch as VT


def sweep_matrices():
    mats = [np.eye(3)]
    for axis in range(3):
        u = np.zeros(3)
        u[axis] = 1.0
        for eps_abs in (0.0, 1e-12, 1e-9, 1e-6, 1e-3, 1e-2, 1e-1):
            for sgn in (1, -1):
                eps = sgn * eps_abs
                for base in (0.0, np.pi / 4, np.pi / 2, np.pi):
                    mats.append(Rotation.from_rotvec(u * (base + eps)).as_matrix())
    rng = np.random.default_rng(0)
    q = rng.normal(size=(10000, 4))
    mats.extend(Rotation.from_quat(q / np.linalg.norm(q, axis=1, keepdims=True)).as_matrix())
    return np.stack(mats)


def check_against_independent(aa, mats):
    ref = Rotation.from_matrix(mats).as_rotvec()
    ang = np.linalg.norm(ref, axis=1)
    err = np.linalg.norm(ref - aa, axis=1)
    err_flip = np.linalg.norm(ref + aa, axis=1)
    ok = (err < 5e-3) | ((np.abs(ang - np.pi) < 1e-4) & (err_flip < 5e-3))
    assert ok.all(), (np.nonzero(~ok)[0][:5], err[~ok][:5])


def test_convert_rotmat_to_axis_angle_sweep():
    mats = sweep_matrices()
    R = torch.from_numpy(mats.astype(np.float32)).requires_grad_(True)
    aa = VT.convert_rotmat_to_axis_angle(R)
    assert not torch.isnan(aa).any()
    check_against_independent(aa.detach().numpy().astype(np.float64), mats)
    # gradient of ||aa|| per matrix is NaN-free (TestVPoser.cpp:36-43)
    aa.norm(dim=1).sum().backward()
    assert not torch.isnan(R.grad).any()


def test_decoder_shapes_and_eval_mode():
    from smplpp_amd.ik import VPoserDecoder as P

    dec = VT.VPoserDecoder(P.synthetic_params())
    assert not dec.net[2].training
    z = np.random.default_rng(1).random((3, 32)).astype(np.float32)
    out, jac = dec.forward_with_jacobian(z)
    assert out.shape == (3, 21, 3) and jac.shape == (3, 63, 32) and np.isfinite(out).all() and np.isfinite(jac).all()
    # the Jacobian is the derivative: finite-difference check on one column
    h = 1e-3
    zp = z.copy()
    zp[:, 5] += h
    zm = z.copy()
    zm[:, 5] -= h
    fd = (dec(torch.from_numpy(zp)).detach().numpy() - dec(torch.from_numpy(zm)).detach().numpy()).reshape(3, 63) / (2 * h)
    # (LeakyReLU kinks crossed inside +-h show up as a few outliers)
    assert np.percentile(np.abs(fd - jac[:, :, 5]), 95) < 2e-3 and np.abs(fd - jac[:, :, 5]).max() < 5e-2


def reference_decoder_golden():
    """The reference's own decoder golden (tests/data/TestVPoser.json, committed as data under tests/golden/) and the gated
    weights it belongs to, when SMPLPP_VPOSER_JSON names a vposer_parameters.json (or the .npz of the same script)."""
    import json
    import os

    import pytest

    path = os.environ.get("SMPLPP_VPOSER_JSON")
    if not path:
        pytest.skip("set SMPLPP_VPOSER_JSON=/path/to/vposer_parameters.json (license-gated VPoser v2 weights, "
                    "scripts/preprocess_vposer.py) to pin the decoder against the reference's golden")
    if not os.path.exists(path):
        pytest.skip("SMPLPP_VPOSER_JSON=%s does not exist" % path)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vposer_reference_golden.json")) as f:
        g = json.load(f)
    return path, np.asarray(g["in"], np.float32), np.asarray(g["out"], np.float32), np.asarray(g["grad"], np.float32)


def test_reference_decoder_golden_restatement():
    """tests/src/TestVPoser.cpp:72-130 on the torch restatement: forward and d||out||/dz at the reference's 1e-6."""
    path, zin, out_gt, grad_gt = reference_decoder_golden()
    from smplpp_amd.ik import VPoserDecoder as P

    if path.endswith(".npz"):
        with np.load(path) as zf:
            params = {k: zf[k] for k in P.KEYS}
    else:
        import json

        with open(path) as f:
            raw = json.load(f)
        params = {k: np.asarray(raw[k], np.float32) for k in P.KEYS}
    dec = VT.VPoserDecoder(params)
    assert not dec.net[2].training  # :129
    z = torch.from_numpy(zin).clone().requires_grad_(True)
    out = dec(z)
    out.norm().backward()
    assert float((out.detach() - torch.from_numpy(out_gt)).norm()) < 1e-6
    assert float((z.grad - torch.from_numpy(grad_gt)).norm()) < 1e-6
