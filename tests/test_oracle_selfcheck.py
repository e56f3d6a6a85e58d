"""CPU: self-validation of the oracle where the reference has no known-answer tests."""
import math

import pytest
import torch

from oracle import corr, covariance, frontend, pgo, se3


def test_volume_is_einsum_and_slices_are_per_query():
    g = torch.Generator().manual_seed(0)
    f1, f2 = torch.randn(2, 16, 3, 5, generator=g), torch.randn(2, 16, 3, 5, generator=g)
    vol = corr.corr_volume(f1, f2, torch.float64)
    assert vol.shape == (2 * 15, 1, 3, 5)
    b, y, x, yy, xx = 1, 2, 3, 0, 4
    assert vol[b * 15 + y * 5 + x, 0, yy, xx].item() == pytest.approx(float((f1[b, :, y, x].double() * f2[b, :, yy, xx].double()).sum()), abs=1e-12)


@pytest.mark.parametrize("r", [1, 3, 4])
def test_lookup_grid_sample_vs_explicit_bilinear(r):
    g = torch.Generator().manual_seed(r)
    B, H, W = 2, 7, 9
    vol = torch.randn(B * H * W, 1, H, W, generator=g)
    coords = corr.coords_grid(B, H, W) + (torch.rand(B, 2, H, W, generator=g) * 2 - 1) * 6
    torch.testing.assert_close(corr.corr_lookup(vol, coords, r), corr.corr_lookup_naive(vol, coords, r), rtol=1e-5, atol=1e-5)


def test_lookup_window_orientation():
    """channel K*i + j samples (x + i - r, y + j - r): the FIRST window index moves along x (RAFT quirk)."""
    H, W, r = 9, 11, 4
    vol = torch.zeros(H * W, 1, H, W)
    q = 4 * W + 5  # query (x=5, y=4)
    vol[q, 0, 4, 7] = 1.0  # slice value at (x=7, y=4) -> dx=+2, dy=0 -> i = r+2, j = r
    out = corr.corr_lookup(vol, corr.coords_grid(1, H, W), r)
    K = 2 * r + 1
    hit = out[0, :, 4, 5].argmax().item()
    assert hit == K * (r + 2) + r and out[0, hit, 4, 5].item() == pytest.approx(1.0, abs=1e-5)


def test_se3_exp_identities():
    g = torch.Generator().manual_seed(1)
    xi = torch.randn(6, generator=g, dtype=torch.float64) * 0.3
    T = se3.se3_exp(xi)
    Ti = se3.se3_inv(T)
    I = se3.se3_mul(T, Ti)
    assert se3.pose_error(I, torch.tensor([0, 0, 0, 0, 0, 0, 1.0], dtype=torch.float64)) < (1e-14, 1e-14)
    assert se3.so3_log(T[3:]).allclose(xi[3:], atol=1e-12)
    p = torch.randn(5, 3, generator=g, dtype=torch.float64)
    R = se3.quat_to_matrix(T[3:])
    assert se3.se3_act(T, p).allclose(p @ R.T + T[:3], atol=1e-13)
    # left update with a tiny step is first-order  T + [rho; phi x]
    small = se3.se3_left_update(T, torch.tensor([1e-9, 0, 0, 0, 0, 0, 123.0], dtype=torch.float64))
    assert (small[:3] - T[:3]).allclose(torch.tensor([1e-9, 0, 0], dtype=torch.float64), atol=1e-15)


@pytest.mark.parametrize("graph", [0, 1, 2])
def test_analytic_jacobian_vs_finite_differences(graph):
    """mirrors AnalyticModule.verify_jacobian (PyposeOptimizers.py:60-73)"""
    prob, _ = pgo.make_synthetic_problem(n=20, seed=3)
    g = pgo._Graph(prob, graph)
    g.T = se3.se3_exp(torch.tensor([0.1, -0.2, 0.05, 0.02, -0.03, 0.01], dtype=torch.float64))
    g.forward()
    J = g.build_jacobian()
    T0 = g.T.clone()
    Jn = torch.zeros_like(J[:, :6])
    for k in range(6):
        d = torch.zeros(7, dtype=torch.float64)
        d[k] = 1e-6
        g.T = se3.se3_left_update(T0, d)
        rp = g.forward().reshape(-1)
        g.T = se3.se3_left_update(T0, -d)
        rm = g.forward().reshape(-1)
        Jn[:, k] = (rp - rm) / 2e-6
    assert (J[:, :6] - Jn).abs().max() / J.abs().max() < 1e-8
    assert J[:, 6].abs().max() == 0


@pytest.mark.parametrize("graph", ["disp", "reproj", "icp"])
def test_lm_recovers_truth_on_exact_observations(graph):
    prob, T_true = pgo.make_synthetic_problem(n=100, seed=6)
    pc = se3.se3_act(se3.se3_inv(T_true), prob.pos_Tw.double())
    prob.pixel2_uv = frontend.point2pixel_NED(pc, prob.K.double()).float()
    prob.pixel2_disp = ((prob.K[0, 0].double() * prob.baseline) / pc[:, 0:1]).float()
    prob.pixel2_d = pc[:, 0:1].float()
    res = pgo.solve(prob, graph)
    dt, dr = se3.pose_error(T_true, res.pose)
    assert dt < 3e-4 and dr < 3e-5  # float32 observations bound the accuracy
    assert res.steps <= 10 and res.history[0][0] >= res.loss


def test_covariance_model_monte_carlo():
    """Covariance_2to3_full vs sampling (as Scripts/AdHoc/CovarianceModel.py does): u, v, d independent Gaussians."""
    g = torch.Generator().manual_seed(0)
    u, v, d = 128.0, 450.0, 10.0
    suu, svv, sdd = 2.0, 5.0, 0.3
    fx = fy = 320.0
    cx, cy = 320.0, 240.0
    n = 400000
    us = u + math.sqrt(suu) * torch.randn(n, generator=g, dtype=torch.float64)
    vs = v + math.sqrt(svv) * torch.randn(n, generator=g, dtype=torch.float64)
    ds = d + math.sqrt(sdd) * torch.randn(n, generator=g, dtype=torch.float64)
    P = torch.stack([ds, (us - cx) * ds / fx, (vs - cy) * ds / fy], -1)
    emp = torch.cov(P.T)
    t = lambda x: torch.tensor([x], dtype=torch.float32)  # noqa: E731
    mod = covariance.Covariance_2to3_full(t(suu), t(0.0), t(svv), t(sdd), t(u), t(v), t(d), fx, fy, cx, cy)[0].double()
    assert torch.allclose(emp, mod, rtol=0.03, atol=2e-3)


def test_huber_and_fast_triggs_definitions():
    x = torch.tensor([0.0, 0.0025, 0.04, 1.0], dtype=torch.float64)  # sqrt: 0, .05, .2, 1
    rho = pgo.huber(x, 0.1)
    assert rho.tolist() == pytest.approx([0.0, 0.0025, 2 * 0.1 * 0.2 - 0.01, 2 * 0.1 * 1.0 - 0.01])
    R = torch.tensor([[0.03, 0.04, 0.0], [0.3, 0.4, 0.0]], dtype=torch.float64)  # norms .05 and .5
    J = torch.ones(6, 7, dtype=torch.float64)
    Rc, Jc = pgo.fast_triggs(R, J, 0.1)
    s = math.sqrt(0.1 / 0.5)
    assert Rc[0].tolist() == pytest.approx(R[0].tolist()) and Rc[1].tolist() == pytest.approx((R[1] * s).tolist())
    assert Jc[:3].eq(1).all() and Jc[3:].allclose(torch.full((3, 7), s, dtype=torch.float64))


def test_local_corr81_matches_index_arithmetic():
    """oracle.corr.local_corr81 (vectorised) vs per-pixel loops written from the reference kernel's index arithmetic."""
    from oracle import corr

    g = torch.Generator().manual_seed(3)
    a, b = torch.randn(2, 5, 6, 7, generator=g), torch.randn(2, 5, 6, 7, generator=g)
    fast = corr.local_corr81(a, b, torch.float64)
    slow = corr.local_corr81_naive(a, b)
    torch.testing.assert_close(fast, slow, rtol=1e-12, atol=1e-12)
    # channel 40 is the zero displacement: mean over C of first * second
    torch.testing.assert_close(fast[:, 40], (a.double() * b.double()).mean(1), rtol=1e-12, atol=1e-12)


def test_patch_embed_oracle_matches_a_direct_convolution_and_its_bf16_twin_stays_close():
    """oracle/patch_embed.py: the F.conv2d chain against an explicit unfold + matmul of the same three layers (independent restatement of stride 2 /
    padding 2 / the bottom-right F.pad), the token layout, and the bf16-operand twin within bf16's operand error of it."""
    import torch.nn.functional as F

    from oracle import patch_embed as ope

    W = ope.make_weights(2)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 1, 60, 80, generator=g) * 8

    def conv_direct(inp, w, b):                       # stride 2, padding 2, via unfold: [S, Cin*36, L] x [Cout, Cin*36]
        S, _, H, Wd = inp.shape
        cols = F.unfold(inp, kernel_size=6, padding=2, stride=2)
        out = torch.einsum("ok,skl->sol", w.reshape(w.shape[0], -1), cols) + b[None, :, None]
        return out.reshape(S, w.shape[0], (H + 4 - 6) // 2 + 1, (Wd + 4 - 6) // 2 + 1)

    y = F.pad(x, (0, 0, 0, 4))                        # 60 -> 64 rows (bottom), 80 columns already a multiple of 8
    y = conv_direct(conv_direct(conv_direct(y, W[0], W[1]).relu(), W[2], W[3]).relu(), W[4], W[5])
    ref = ope.patch_embed_proj(x, *W)
    assert ref.shape == (3, 64, 8, 10)
    torch.testing.assert_close(ref, y, rtol=1e-4, atol=1e-5)
    tok = ope.to_tokens(ref)
    assert tok.shape == (3, 80, 64) and torch.equal(tok[1, 23], ref[1, :, 2, 3])
    bf = ope.patch_embed_proj_bf16(x, *W)
    assert (bf - ref).abs().max() <= 2e-2 * ref.abs().max()
    assert (bf - ref).abs().max() > 0                 # it IS a different arithmetic
    hf = ope.patch_embed_proj_f16(x, *W)
    assert 0 < (hf - ref).abs().max() <= 2.5e-3 * ref.abs().max() and (hf - ref).abs().max() < (bf - ref).abs().max()
