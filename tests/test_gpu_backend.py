"""GPU parity: selectors (bit-exact indices), covariance model, tracking gathers, epilogue, PGO vs the CPU oracle."""
import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------- selectors (bit-exact)
@pytest.mark.parametrize("H,W,seed,nan_frac", [(480, 640, 2, 0.0), (480, 640, 7, 0.01), (96, 130, 3, 0.02), (720, 1280, 4, 0.0)])
def test_selector_nodepth_bit_exact(gpu, H, W, seed, nan_frac):
    from macvo_amd import ops
    from oracle import selector

    fc = synth.flow_cov_maps(H, W, seed, nan_frac)
    torch.manual_seed(1234)
    ref_px, ref_cand, aux = selector.cov_aware_selector_nodepth(fc.clone(), 200, 7, 32, 100.0)
    torch.manual_seed(1234)
    cands = ops.kp_select("nodepth", H, W, flow_cov=fc.to(gpu), kernel_size=7, mask_width=32, max_match_cov=100.0)
    px = cands.finish(200)
    assert cands.n == ref_cand.shape[0]
    assert torch.equal(cands.candidates_vu().cpu(), ref_cand)
    assert px.dtype == torch.int64 and torch.equal(px.cpu(), ref_px)
    st = cands.stats.cpu()
    assert st[0].item() == pytest.approx(aux["median"], rel=0, abs=0)
    assert int(cands.count[1].item()) == aux["n_nms"]


@pytest.mark.parametrize("H,W,cov_is_log,nan_frac,masked", [(480, 640, True, 0.0, False), (480, 640, True, 0.01, True),
                                                             (96, 130, False, 0.02, False), (720, 1280, True, 0.0, False)])
def test_fused_epilogue_selector_is_bitwise_the_two_calls(gpu, H, W, cov_is_log, nan_frac, masked):
    """mv_frontend_epilogue_select_lanes (what the frame driver launches) against frontend_epilogue followed by kp_select: every
    map, the candidate list, its count, the NMS count and the median must agree bit for bit — including NaNs in the covariance
    planes, an image whose size is no multiple of the 64 x 16 tile, the already-exponentiated input form and selector masks."""
    from macvo_amd import ops

    g = torch.Generator().manual_seed(77)
    flow = torch.randn(2, 2, H, W, generator=g) * 4
    cov = torch.randn(2, 2, H, W, generator=g) * 0.5
    if not cov_is_log:
        cov = torch.exp(2 * cov)
    if nan_frac:
        cov[torch.rand(cov.shape, generator=g) < nan_frac] = float("nan")
    ma = (torch.rand(1, 1, H, W, generator=g) > 0.3) if masked else None
    flow, cov = flow.to(gpu), cov.to(gpu)
    ma_d = ma.to(gpu) if masked else None
    maps = ops.frontend_epilogue(flow, cov, 0.25, 320.0, cov_is_log=cov_is_log, enforce_positive_disparity=True)
    cands = ops.kp_select("nodepth", H, W, flow_cov=maps.flow_cov, kernel_size=7, mask_width=32, max_match_cov=100.0, mask_a=ma_d)
    fmaps, fcands = ops.frontend_epilogue_select(flow, cov, 0.25, 320.0, cov_is_log=cov_is_log, enforce_positive_disparity=True,
                                                 kernel_size=7, mask_width=32, max_match_cov=100.0, mask_a=ma_d)
    for name in ("depth", "depth_cov", "disparity", "disparity_cov", "flow", "flow_cov"):
        a, b = getattr(maps, name), getattr(fmaps, name)
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), name
    assert torch.equal(maps.bad_mask, fmaps.bad_mask)
    assert cands.n == fcands.n and cands.n > 0
    assert torch.equal(cands.cand[: cands.n], fcands.cand[: fcands.n])
    assert torch.equal(cands.count, fcands.count)
    assert torch.equal(cands.stats.view(torch.int32), fcands.stats.view(torch.int32))


def test_selector_nodepth_plateau_and_mask(gpu):
    """Plateaus (equality NMS keeps every tied pixel), a validity mask, and a low max_match_cov cap."""
    from macvo_amd import ops
    from oracle import selector

    H, W = 128, 192
    fc = synth.flow_cov_maps(H, W, 11)
    fc[:, :2, 40:60, 50:90] = 0.01          # constant plateau = many ties at the minimum
    g = torch.Generator().manual_seed(5)
    mask = torch.rand(1, 1, H, W, generator=g) > 0.3
    torch.manual_seed(99)
    ref_px, ref_cand, _ = selector.cov_aware_selector_nodepth(fc.clone(), 50, 5, 16, 0.5, match_mask=mask)
    torch.manual_seed(99)
    c = ops.kp_select("nodepth", H, W, flow_cov=fc.to(gpu), mask_b=mask.to(gpu), kernel_size=5, mask_width=16, max_match_cov=0.5)
    assert torch.equal(c.candidates_vu().cpu(), ref_cand)
    assert torch.equal(c.finish(50).cpu(), ref_px)


@pytest.mark.parametrize("H,W,with_flow", [(480, 640, True), (480, 640, False), (100, 140, True)])
def test_selector_full_bit_exact(gpu, H, W, with_flow):
    from macvo_amd import ops
    from oracle import selector

    fc = synth.flow_cov_maps(H, W, 2, 0.005) if with_flow else None
    d0, d0c = synth.depth_maps(H, W, 3)
    d1, d1c = synth.depth_maps(H, W, 4)
    d0c[0, 0, 5, 7] = float("nan")
    max_depth = 320.0 * 0.25
    torch.manual_seed(4321)
    ref_px, ref_cand, aux = selector.cov_aware_selector(d0, d0c, d1, d1c, None if fc is None else fc.clone(), 200,
                                                        max_depth, 7, 32, 250.0, 100.0)
    torch.manual_seed(4321)
    c = ops.kp_select("full", H, W, flow_cov=None if fc is None else fc.to(gpu), depth0=d0.to(gpu), depth0_cov=d0c.to(gpu),
                      depth1=d1.to(gpu), depth1_cov=d1c.to(gpu), kernel_size=7, mask_width=32, max_depth=max_depth,
                      max_depth_cov=250.0, max_match_cov=100.0)
    assert torch.equal(c.candidates_vu().cpu(), ref_cand)
    assert torch.equal(c.finish(200).cpu(), ref_px)
    assert c.stats[2].item() == aux["median_depth_cov"]


def test_selector_mapping_bit_exact(gpu):
    from macvo_amd import ops
    from oracle import selector

    H, W = 480, 640
    d0, d0c = synth.depth_maps(H, W, 3)
    torch.manual_seed(7)
    ref_px, ref_cand, _ = selector.mapping_point_selector(d0, d0c, 2000, 20.0, 0.2, 32)
    assert ref_cand.shape[0] > 2000
    torch.manual_seed(7)
    c = ops.kp_select("mapping", H, W, depth0=d0.to(gpu), depth0_cov=d0c.to(gpu), mask_width=32, max_depth=20.0, max_depth_cov=0.2)
    assert torch.equal(c.candidates_vu().cpu(), ref_cand)
    assert torch.equal(c.finish(2000).cpu(), ref_px)


# ------------------------------------------------------------------------------- covariance model
@pytest.mark.parametrize("n,float_kp", [(200, False), (2000, False), (37, True)])
def test_match_cov(gpu, n, float_kp):
    from macvo_amd import ops
    from oracle import covariance

    H, W = 480, 640
    depth, _ = synth.depth_maps(H, W, 3)
    kp = synth.keypoints(n, H, W, 5)
    g = torch.Generator().manual_seed(8)
    if float_kp:
        kp = kp.float() + torch.rand(n, 2, generator=g)
    fcov = torch.exp(2 * 0.5 * torch.randn(n, 3, generator=g))
    fcov[:, 2] = 0.2 * torch.randn(n, generator=g) * fcov[:, :2].min(dim=1).values   # PD off-diagonal
    fcov[:5, 0] = 0.01  # below the 0.0625 clamp
    fcov[:5, 2] = 0.0
    fcov[5, :] = torch.tensor([0.3, 0.3, 0.5])  # indefinite Sigma (det < 0): the reference yields NaN, so must we
    K = (320.0, 320.0, 320.0, 240.0)
    fc_ref = fcov.clone()
    ref, aux = covariance.match_covariance(kp, depth, None, fc_ref, *K, return_aux=True)
    fc_dev = fcov.to(gpu)
    R = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))[0]
    out, out_rot, stats = ops.match_cov(depth.to(gpu), kp.to(gpu), fc_dev, None, *K, rot=R, want_stats=True)
    assert out.dtype == torch.float64
    # in-place clamp of the caller's flow_cov, bit-exact
    assert torch.equal(fc_dev.cpu(), fc_ref)
    assert ref[5].isnan().any() and out[5].isnan().any()
    torch.testing.assert_close(stats[:, 0].cpu(), aux["wavg"], rtol=2e-5, atol=1e-5, equal_nan=True)
    torch.testing.assert_close(stats[:, 1].cpu(), aux["wvar"], rtol=2e-4, atol=1e-6, equal_nan=True)
    torch.testing.assert_close(out.cpu(), ref, rtol=2e-4, atol=1e-7, equal_nan=True)
    torch.testing.assert_close(out_rot.cpu(), covariance.rotate_covariance(R, out.cpu()), rtol=1e-12, atol=1e-14, equal_nan=True)


def test_match_cov_pair_equals_two_calls(gpu):
    from macvo_amd import ops

    H, W, n = 240, 320, 150
    d0, _ = synth.depth_maps(H, W, 3)
    d1, _ = synth.depth_maps(H, W, 4)
    g = torch.Generator().manual_seed(3)
    k0 = synth.keypoints(n, H, W, 5).float()
    k1 = k0 + torch.rand(n, 2, generator=g) * 3
    s0 = torch.ones(n, 3) * 0.25
    s0[:, 2] = 0
    s1 = torch.rand(n, 3, generator=g) * 0.5
    s1[:, 2] = 0
    K = (160.0, 160.0, 160.0, 120.0)
    R = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))[0].to(gpu)
    a0, a1 = s0.clone().to(gpu), s1.clone().to(gpu)
    c0, c0w = ops.match_cov(d0.to(gpu), k0.to(gpu), a0, None, *K, rot=R)
    c1 = ops.match_cov(d1.to(gpu), k1.to(gpu), a1, None, *K)
    b0, b1 = s0.clone().to(gpu), s1.clone().to(gpu)
    p0, p0w, p1 = ops.match_cov_pair(d0.to(gpu), k0.to(gpu), b0, d1.to(gpu), k1.to(gpu), b1, *K, rot=R)
    assert torch.equal(p0, c0) and torch.equal(p0w, c0w) and torch.equal(p1, c1)
    assert torch.equal(a0, b0) and torch.equal(a1, b1)


def test_pose_apply_is_bitwise_backproject_plus_rotated_covariance(gpu):
    """mv_pose_apply_lanes (the pose-dependent remainder the frame driver runs behind the previous solve) against the kernels that
    used to produce the same tables: mv_backproject with a pose (pos_Tw, rot) and mv_match_cov with rot (R cov R^T)."""
    from macvo_amd import ops

    g = torch.Generator().manual_seed(21)
    n, H, W = 200, 480, 640
    depth = (torch.rand(1, 1, H, W, generator=g) * 40 + 2).to(gpu)
    kp = torch.stack([torch.randint(40, W - 40, (n,), generator=g), torch.randint(40, H - 40, (n,), generator=g)], 1).float().to(gpu)
    dvals = (torch.rand(n, generator=g) * 30 + 1).to(gpu)
    q = torch.randn(4, generator=g)
    pose = torch.cat([torch.randn(3, generator=g), q / q.norm()]).float().to(gpu)
    K4 = (320.0, 320.0, 320.0, 240.0)
    pos_Tc, pos_Tw, rot = ops.backproject(kp, dvals, K4, pose, want_rot=True)
    sigma = (torch.rand(n, 3, generator=g) * torch.tensor([2.0, 2.0, 0.2])).float().to(gpu)
    cov, cov_w = ops.match_cov(depth, kp, sigma.clone(), None, *K4, rot=rot.view(3, 3))
    pos_Tw2, rot2, cov_w2 = ops.pose_apply(pose, pos_Tc, cov)
    assert torch.equal(pos_Tw.view(torch.int32), pos_Tw2.view(torch.int32))
    assert torch.equal(rot.reshape(9).view(torch.int64), rot2.view(torch.int64))
    assert torch.equal(cov_w.view(torch.int64), cov_w2.view(torch.int64))


def test_match_cov_given_depth_cov_and_nan(gpu):
    """depth_cov branch (flow_cov None in the reference == default sigma, use given variance) and NaN propagation."""
    from macvo_amd import ops
    from oracle import covariance

    H, W, n = 200, 300, 64
    depth, dcov = synth.depth_maps(H, W, 3)
    kp = synth.keypoints(n, H, W, 6)
    K = (320.0, 320.0, 150.0, 100.0)
    dc = dcov[0, 0, kp[:, 1], kp[:, 0]].contiguous()
    ref = covariance.match_covariance(kp, depth, dc, None, *K)
    fc = torch.ones(n, 3) * 0.25
    fc[:, 2] = 0
    out = ops.match_cov(depth.to(gpu), kp.to(gpu), fc.to(gpu), dc.to(gpu), *K, use_patch_var=False)
    torch.testing.assert_close(out.cpu(), ref, rtol=2e-4, atol=1e-8)
    depth2 = depth.clone()
    depth2[0, 0, kp[0, 1], kp[0, 0]] = float("nan")
    out2 = ops.match_cov(depth2.to(gpu), kp.to(gpu), fc.to(gpu), None, *K).cpu()
    assert out2[0].isnan().any() and not out2[-1].isnan().any()


# ------------------------------------------------------------------------------- epilogue + tracking
def test_frontend_epilogue_and_track(gpu):
    from macvo_amd import ops
    from oracle import frontend

    H, W = 480, 640
    g = torch.Generator().manual_seed(21)
    flow = torch.randn(2, 2, H, W, generator=g) * 6
    flow[0, 0] = -(torch.rand(H, W, generator=g) * 40 + 2)
    logcov = torch.randn(2, 2, H, W, generator=g) * 0.5
    bl, fx = 0.25, 320.0
    m = ops.frontend_epilogue(flow.to(gpu), logcov.to(gpu), bl, fx, cov_is_log=True, enforce_positive_disparity=True)
    cov = torch.exp(logcov * 2)
    depth, depth_cov, disp, disp_cov, bad = frontend.inference_2_depth(flow[0:1], cov[0:1], bl, fx, True)
    assert torch.equal(m.disparity.cpu(), disp)
    torch.testing.assert_close(m.disparity_cov.cpu(), disp_cov, rtol=3e-6, atol=0)      # expf ulp differences only
    assert torch.equal(m.depth.cpu(), depth)
    torch.testing.assert_close(m.depth_cov.cpu(), depth_cov, rtol=4e-6, atol=0)
    assert torch.equal(m.bad_mask.cpu(), bad)
    assert torch.equal(m.flow.cpu(), flow[1:2])
    torch.testing.assert_close(m.flow_cov.cpu(), frontend.from_partial_cov(cov[1:2]), rtol=3e-6, atol=0)

    # exact-cov variant for bit-exact depth_cov: feed sigma^2 directly
    m2 = ops.frontend_epilogue(flow.to(gpu), cov.to(gpu), bl, fx, cov_is_log=False)
    assert torch.equal(m2.depth_cov.cpu(), depth_cov) and torch.equal(m2.flow_cov.cpu(), frontend.from_partial_cov(cov[1:2]))

    kp0 = synth.keypoints(300, H, W, 9, border=32)
    d0 = {"depth": depth, "disparity": disp, "disparity_uncertainty": disp_cov, "cov": depth_cov}
    fc3 = frontend.from_partial_cov(cov[1:2])
    ref = frontend.track_keypoints(kp0, flow[1:2], fc3, d0, d0, 32, H, W, 0.25)
    tr = ops.kp_track(kp0.to(gpu), m2.flow, m2.flow_cov, m2, m2, 32)
    inb = tr.inbound.cpu()
    assert torch.equal(inb, ref["inbound_mask"])
    assert 0 < int(inb.sum()) < 300
    assert torch.equal(tr.kp1_uv.cpu()[inb], ref["kp1_uv"])
    v = tr.vals.cpu().T[inb]
    assert torch.equal(v[:, 0], ref["kp0_d"]) and torch.equal(v[:, 4], ref["kp1_d"])
    assert torch.equal(v[:, 1], ref["kp0_disparity"][0]) and torch.equal(v[:, 5], ref["kp1_disparity"][0])
    assert torch.equal(v[:, 2], ref["kp0_sigma_disparity"][0]) and torch.equal(v[:, 6], ref["kp1_sigma_disparity"][0])
    assert torch.equal(v[:, 3], ref["kp0_sigma_dd"]) and torch.equal(v[:, 7], ref["kp1_sigma_dd"])
    assert torch.equal(v[:, 8:11], ref["kp1_sigma_uv"])
    assert torch.equal(tr.sigma1.cpu()[inb], ref["kp1_sigma_uv"]) and torch.equal(tr.sigma0.cpu()[inb], ref["kp0_sigma_uv"])
    assert torch.equal(tr.kp0_uv.cpu()[inb], ref["kp0_uv"].float())


# ------------------------------------------------------------------------------- PGO
from tools.synth import pgo_batch as _to_batch  # noqa: E402  (the tests keep the old name)


@pytest.mark.parametrize("graph", ["disp", "reproj", "icp"])
def test_pgo_matches_oracle(gpu, graph):
    """Pose within 1e-4 m / 1e-4 rad of the oracle after the same iteration count (north_star tolerance);
    in practice fp64-roundoff close."""
    from macvo_amd import ops
    from oracle import pgo, se3

    cases = [dict(n=200, seed=6), dict(n=200, seed=7, outlier_frac=0.1), dict(n=37, seed=8), dict(n=12, seed=9),
             dict(n=500, seed=10, trans_sigma=0.4, rot_sigma=0.08), dict(n=64, seed=11, outlier_frac=0.3)]
    probs = [pgo.make_synthetic_problem(**c)[0] for c in cases]
    pose, info = ops.pgo_solve(_to_batch(probs, gpu), graph)
    pose, info = pose.cpu(), info.cpu()
    for k, p in enumerate(probs):
        ref = pgo.solve(p, graph)
        dt, dr = se3.pose_error(ref.pose, pose[k])
        assert dt <= 1e-4 and dr <= 1e-4, (k, dt, dr)
        assert dt <= 1e-8 and dr <= 1e-8, (k, dt, dr)       # actual agreement is ~fp64 roundoff
        assert int(info[k, 1]) == ref.steps, (k, info[k], ref.steps)
        assert info[k, 0].item() == pytest.approx(ref.loss, rel=1e-8, abs=1e-12)


@pytest.mark.parametrize("nprob_pad", [0, 520])
@pytest.mark.parametrize("graph", ["disp", "reproj", "icp"])
def test_pgo_kernel_equals_host_twin(gpu, graph, nprob_pad):
    """mv_pgo_solve against tests/c_abi/pgo_twin.cpp — the same arithmetic header (csrc/pgo_math.h) and the same reduction trees
    replayed lane by lane on the host: LM steps and reject counts equal, pose / loss equal to fp64 roundoff (the device's rsqrt is the
    one operation the host rounds differently).  nprob_pad = 520 pushes the batch over the 512-problem switch to the one-wave
    throughput variant (the twin follows with nw = 1)."""
    from macvo_amd import ops
    from oracle import pgo
    from tests import pgo_twin

    cases = [dict(n=200, seed=6), dict(n=200, seed=7, outlier_frac=0.1), dict(n=37, seed=8), dict(n=12, seed=9),
             dict(n=500, seed=10, trans_sigma=0.4, rot_sigma=0.08), dict(n=64, seed=11, outlier_frac=0.3), dict(n=256, seed=1),
             dict(n=150, seed=21, outlier_frac=0.2, trans_sigma=0.3, rot_sigma=0.05)]
    probs = [pgo.make_synthetic_problem(**c)[0] for c in cases]
    probs += [pgo.make_synthetic_problem(n=20 + (k % 7), seed=100 + k)[0] for k in range(nprob_pad)]
    pose, info = ops.pgo_solve(_to_batch(probs, gpu), graph)
    pose_t, info_t = pgo_twin.solve(_to_batch(probs, torch.device("cpu")), graph)
    pose, info = pose.cpu(), info.cpu()
    assert torch.equal(info[:, 1:3], info_t[:, 1:3]), (info[:8], info_t[:8])
    torch.testing.assert_close(pose, pose_t, rtol=0, atol=1e-11)
    torch.testing.assert_close(info[:, 0], info_t[:, 0], rtol=1e-11, atol=1e-13)


def test_pgo_recovers_truth_noise_free(gpu):
    from macvo_amd import ops
    from oracle import pgo, se3

    prob, T_true = pgo.make_synthetic_problem(n=200, seed=6)
    # make observations exact
    pc = se3.se3_act(se3.se3_inv(T_true), prob.pos_Tw.double())
    from oracle.frontend import point2pixel_NED
    prob.pixel2_uv = point2pixel_NED(pc, prob.K.double()).float()
    prob.pixel2_disp = ((prob.K[0, 0].double() * prob.baseline) / pc[:, 0:1]).float()
    pose, info = ops.pgo_solve(_to_batch([prob], gpu), "disp")
    dt, dr = se3.pose_error(T_true, pose[0].cpu())
    assert dt < 2e-4 and dr < 2e-5   # limited by the float32 observations, not by the solver
