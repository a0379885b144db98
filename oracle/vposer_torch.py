"""TEST INFRASTRUCTURE ONLY — op-for-op restatement of /root/reference/src/VPoser.cpp in Python torch (fp32, CPU),
differentiated by torch autograd exactly as the reference differentiates it (node/node.cpp:761-772).  src/VPoser.cpp
itself cannot be compiled here (xtensor / nlohmann-json in loadParamsFromJson).

Pinned by the reference's self-contained sweep tests/src/TestVPoser.cpp:16-70 (convertRotMatToAxisAngle against an
independent rotation->axis-angle, tol 5e-3, NaN-free gradients) — see tests/test_oracle_vposer.py.  The decoder golden
tests/data/TestVPoser.json needs the license-gated vposer_parameters.json and cannot be reproduced: the MLP part is
"parity unpinned" (it is three torch.nn.Linear + LeakyReLU; the synthetic weights follow torch's default init).
"""
from __future__ import annotations

import math

import numpy as np
import torch

EPS = float(np.finfo(np.float32).eps)  # std::numeric_limits<float>::epsilon()  (:33)
EPS_SQRT = math.sqrt(EPS)  # :34
EPS_SQRT2 = math.sqrt(EPS_SQRT)  # :35


def convert_rotmat_to_axis_angle(rotMat: torch.Tensor) -> torch.Tensor:
    """src/VPoser.cpp:25-120, same masks, same in-place writes."""
    trace = rotMat.diagonal(0, 1, 2).sum(-1)  # :37
    theta = torch.arccos((1.0 - EPS) * 0.5 * (trace - 1.0))  # :41
    w = torch.stack([rotMat[:, 2, 1] - rotMat[:, 1, 2], rotMat[:, 0, 2] - rotMat[:, 2, 0], rotMat[:, 1, 0] - rotMat[:, 0, 1]], dim=1)
    aa = torch.empty_like(w)
    pi_c = (1.0 + trace) < EPS_SQRT2  # :53
    if pi_c.any():
        R = rotMat[pi_c]
        tr = trace[pi_c]
        s = (2.0 * R.diagonal(0, 1, 2) + (1.0 - tr).view(-1, 1).expand(-1, 3)) / (3.0 - tr).view(-1, 1)  # :54-56
        tn2 = torch.sqrt(s + EPS) * theta[pi_c].view(-1, 1)  # :60
        c1 = theta[pi_c] > math.pi - 1e-4  # :62
        sign = torch.ones_like(tn2)
        # :64-94 — sign fixes near pi
        y1 = c1 & (tn2[:, 0] > 0.0)
        sign[:, 1] = torch.where(y1 & ((R[:, 0, 1] + R[:, 1, 0]) < 0.0), -sign[:, 1], sign[:, 1])
        sign[:, 2] = torch.where(y1 & ((R[:, 0, 2] + R[:, 2, 0]) < 0.0), -sign[:, 2], sign[:, 2])
        n1 = c1 & ~(tn2[:, 0] > 0.0) & (tn2[:, 1] > 0.0)
        sign[:, 2] = torch.where(n1 & ((R[:, 1, 2] + R[:, 2, 1]) < 0.0), -sign[:, 2], sign[:, 2])
        # :96-99 — away from pi: follow the sign of w
        wn = w[pi_c]
        sign = torch.where((~c1).view(-1, 1) & ~(wn >= 0.0), -sign, sign)
        aa[pi_c] = tn2 * sign
    if (~pi_c).any():
        wn = w[~pi_c]
        th = theta[~pi_c]
        zero_c = torch.abs(3.0 - trace[~pi_c]) < EPS_SQRT  # :105
        out = torch.empty_like(wn)
        if zero_c.any():
            tz = th[zero_c]
            out[zero_c] = 0.5 * wn[zero_c] * (1.0 + tz**2 / 6.0 + tz**4 * 7.0 / 360.0).view(-1, 1)  # :107-111
        if (~zero_c).any():
            tnz = th[~zero_c]
            out[~zero_c] = wn[~zero_c] * torch.div(tnz, 2.0 * torch.sin(tnz)).view(-1, 1)  # :112-116
        aa[~pi_c] = out
    return aa


def continuous_rot_repr_decoder(x: torch.Tensor) -> torch.Tensor:
    """ContinousRotReprDecoderImpl::forward (:129-141)."""
    r = x.view(-1, 3, 2)
    col1, col2 = r[:, :, 0], r[:, :, 1]
    axis1 = torch.nn.functional.normalize(col1, dim=1)
    axis2 = torch.nn.functional.normalize(col2 - (axis1 * col2).sum(1, True) * axis1, dim=-1)
    axis3 = torch.cross(axis1, axis2, dim=1)
    return torch.stack([axis1, axis2, axis3], dim=-1).view(-1, 3, 3)


class VPoserDecoder(torch.nn.Module):
    """VPoserDecoderImpl (:143-167): Sequential(Linear, LeakyReLU, Dropout(0.1), Linear, LeakyReLU, Linear, 6D->R)."""

    def __init__(self, params):
        super().__init__()
        self.net = torch.nn.Sequential(
            torch.nn.Linear(32, 512), torch.nn.LeakyReLU(), torch.nn.Dropout(0.1), torch.nn.Linear(512, 512),
            torch.nn.LeakyReLU(), torch.nn.Linear(512, 126))
        with torch.no_grad():
            for idx, key in ((0, "decoder_net.0"), (3, "decoder_net.3"), (5, "decoder_net.5")):
                self.net[idx].weight.copy_(torch.from_numpy(np.asarray(params[key + ".weight"], np.float32)))
                self.net[idx].bias.copy_(torch.from_numpy(np.asarray(params[key + ".bias"], np.float32)))
        self.eval()  # the reference asserts Dropout is in eval (tests/src/TestVPoser.cpp:129)

    def forward(self, latent):
        n = latent.shape[0]
        return convert_rotmat_to_axis_angle(continuous_rot_repr_decoder(self.net(latent))).view(n, -1, 3)

    def forward_with_jacobian(self, z: np.ndarray):
        """out [n,21,3], d(out)/dz [n,63,32] by one backward() per output row (what node.cpp's autograd provides)."""
        zt = torch.from_numpy(np.asarray(z, np.float32)).clone().requires_grad_(True)
        out = self.forward(zt)
        n = zt.shape[0]
        jac = np.zeros((n, 63, 32), np.float32)
        flat = out.reshape(n, 63)
        for r in range(63):
            g, = torch.autograd.grad(flat[:, r].sum(), zt, retain_graph=True)
            jac[:, r, :] = g.numpy()
        return out.detach().numpy(), jac
