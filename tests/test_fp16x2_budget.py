"""Error budget of the fp16x2 operand scheme of the default fused FK kernel (smplpp_amd/csrc/skin_h.hip), on the CPU.

Every fp32 operand x of the blend-shape GEMM rest = T + S.beta + P.c (/root/reference/src/BlendShape.cpp:670-683, 762-765)
is carried as two fp16 pieces of s.x (s a power of two): hi = fp16(s x), lo = fp16(s x - hi); a product is evaluated as
hi.hi + hi.lo + lo.hi in fp32 MFMA accumulators. This test restates the split in numpy and bounds the REPRESENTATION error
(exact accumulation) at real-SMPL magnitudes, with and without fp16 subnormals (a matrix pipe that flushed them would
still be inside the bar), against the 1e-5 m parity bar. The GPU-side counterpart is
tests/test_fk_gpu.py::test_fk_fp16x2_at_real_smpl_magnitudes."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation


def _split(x, s, ftz):
    xs = (x.astype(np.float32) * np.float32(s)).astype(np.float32)
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float32)).astype(np.float16)
    if ftz:
        tiny = np.float16(6.1035e-5)
        hi = np.where(np.abs(hi) < tiny, np.float16(0), hi)
        lo = np.where(np.abs(lo) < tiny, np.float16(0), lo)
    assert np.isfinite(hi.astype(np.float32)).all()
    return hi.astype(np.float64), lo.astype(np.float64)


@pytest.mark.parametrize("name,pmag,bmag,th", [("synthetic", 0.002, 1.0, 0.3), ("real", 0.05 / 3, 3.0, 1.0), ("extreme", 0.05, 3.0, 1.5)])
@pytest.mark.parametrize("ftz", [False, True])
def test_fp16x2_three_product_error(name, pmag, bmag, th, ftz):
    rng = np.random.default_rng(0)
    N, K, C = 128, 218, 2000
    A = np.zeros((N, K), np.float32)
    R = Rotation.from_rotvec(rng.normal(0, th, (N * 23, 3))).as_matrix().reshape(N, 23, 9) - np.eye(3).reshape(9)
    A[:, :207] = R.reshape(N, 207)
    A[:, 207:217] = rng.normal(0, 1, (N, 10)) * bmag
    A[:, 217] = 1
    B = np.zeros((K, C), np.float32)
    B[:207] = rng.normal(0, pmag, (207, C)).clip(-3 * pmag, 3 * pmag)
    B[207:217] = rng.normal(0, 0.03, (10, C)) * (1 / np.arange(1, 11))[:, None]
    B[217] = rng.uniform(-0.9, 0.9, C)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    sA = 64.0  # HB_SA in smplpp_amd/csrc/common.h
    sB = 2.0 ** np.floor(np.log2(32768 / np.abs(B).max()))  # smplpp_model_create
    ah, al = _split(A, sA, ftz)
    bh, bl = _split(B, sB, ftz)
    got = (al @ bh + ah @ bl + ah @ bh) / (sA * sB)
    err = np.abs(got - ref).max()
    assert err < 6e-7, (name, ftz, err)  # 16x inside the 1e-5 m bar; a plain fp32 GEMM of the same data is at 1e-7..2e-6
