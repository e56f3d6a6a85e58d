"""CPU: the split cost-volume arithmetic (oracle/corr_split.py) stays inside the fp32 parity bar of the exact path —
|out - einsum_f64| <= 2e-5 sqrt(C) for N(0,1) features — and its per-row power-of-two scaling is exact."""
import numpy as np
import pytest

from oracle import corr_split


def _feats(n1, n2, c, seed):
    g = np.random.default_rng(seed)
    return g.standard_normal((n1, c)).astype(np.float32), g.standard_normal((n2, c)).astype(np.float32)


@pytest.mark.parametrize("mode", ["f16x2", "bf16x3"])
def test_split_arithmetic_is_inside_the_fp32_bar(mode):
    C = 256
    f1, f2 = _feats(300, 384, C, 0)
    ref = f1.astype(np.float64) @ f2.astype(np.float64).T
    out = corr_split.corr_volume_split(f1, f2, mode)
    assert out.dtype == np.float32
    err = np.abs(out.astype(np.float64) - ref).max()
    assert err <= 2e-5 * C ** 0.5, err
    f32 = np.abs((f1 @ f2.T).astype(np.float64) - ref).max()      # a float32 matmul on the CPU: the same class of error
    assert err <= 4 * f32 + 1e-6


def test_f16x2_pieces_and_scales():
    f, _ = _feats(200, 1, 256, 3)
    f[5] *= 2.0 ** 40
    f[6] *= 2.0 ** -40
    f[7] = 0.0
    h0, h1, sh = corr_split.pack_f16x2(f)
    assert sh[7] == 0 and np.all(np.isfinite(h0.astype(np.float32))) and np.all(np.isfinite(h1.astype(np.float32)))
    m = np.abs(np.ldexp(f, sh[:, None])).max(axis=1)
    nz = np.abs(f).max(axis=1) > 0
    assert np.all((m[nz] >= 2.0 ** 14) & (m[nz] < 2.0 ** 15))
    back = np.ldexp(h0.astype(np.float64) + h1.astype(np.float64), -sh[:, None])
    tol = np.abs(f.astype(np.float64)) * 2.0 ** -21 + np.abs(f).max(axis=1, keepdims=True).astype(np.float64) * 2.0 ** -39
    assert np.all(np.abs(back - f.astype(np.float64)) <= tol)


@pytest.mark.parametrize("span", [40, 100])
def test_f16x2_dynamic_range_rows(span):
    """rows spread over 2^+-40 / 2^+-100 (most of the fp32 range): the error relative to sum |a||b| stays at the fp32 level (the per-row
    scales carry the range; they are not clamped)"""
    C = 256
    f1, f2 = _feats(128, 128, C, 7)
    g = np.random.default_rng(1)
    e1 = g.integers(-span, span + 1, size=(128, 1))
    f1 *= (2.0 ** e1).astype(np.float32)
    f2 *= (2.0 ** np.clip(g.integers(-span, span + 1, size=(128, 1)), -110 - e1.min(), 110 - e1.max())).astype(np.float32)   # products stay finite
    ref = f1.astype(np.float64) @ f2.astype(np.float64).T
    mag = np.abs(f1).astype(np.float64) @ np.abs(f2).astype(np.float64).T
    out = corr_split.corr_volume_split(f1, f2, "f16x2").astype(np.float64)
    assert np.all(np.abs(out - ref) <= 4e-7 * mag + 1e-300)
