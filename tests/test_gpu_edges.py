"""GPU: edge cases — empty / degenerate inputs, NaNs, lost track, ragged sizes (the situations the reference guards
against in Odometry/MACVO.py:303-307, OutlierFilter.py:91-100, KeypointSelector.py border/NaN handling)."""
import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu


def test_selector_all_nan_and_zero_mask_width(gpu):
    from macvo_amd import ops
    from oracle import selector

    H, W = 64, 96
    fc = torch.full((1, 3, H, W), float("nan"))
    c = ops.kp_select("nodepth", H, W, flow_cov=fc.to(gpu), kernel_size=7, mask_width=8, max_match_cov=100.0)
    assert c.n == 0 and c.finish(200).shape == (0, 2) and int(c.count[1].item()) == 0
    assert torch.isnan(c.stats[0]).item()
    # mask_width = 0: the reference's `[..., 0:-0, 0:-0] = True` selects nothing (KeypointSelector.py:382-385)
    fc = synth.flow_cov_maps(H, W, 2)
    torch.manual_seed(0)
    ref_px, ref_cand, _ = selector.cov_aware_selector_nodepth(fc.clone(), 50, 7, 0, 100.0)
    assert ref_cand.shape[0] == 0
    c = ops.kp_select("nodepth", H, W, flow_cov=fc.to(gpu), kernel_size=7, mask_width=0, max_match_cov=100.0)
    assert c.n == 0


def test_selector_fewer_candidates_than_requested_and_kernel_size_1(gpu):
    from macvo_amd import ops
    from oracle import selector

    H, W = 72, 100
    fc = synth.flow_cov_maps(H, W, 3)
    for ks, npt in ((15, 500), (1, 30)):
        torch.manual_seed(3)
        ref_px, ref_cand, _ = selector.cov_aware_selector_nodepth(fc.clone(), npt, ks, 10, 100.0)
        torch.manual_seed(3)
        c = ops.kp_select("nodepth", H, W, flow_cov=fc.to(gpu), kernel_size=ks, mask_width=10, max_match_cov=100.0)
        px = c.finish(npt)
        assert torch.equal(px.cpu(), ref_px) and px.shape[0] == min(npt, ref_cand.shape[0])


def test_zero_keypoints_through_backend_ops(gpu):
    from macvo_amd import ops

    H, W = 64, 96
    depth = synth.depth_maps(H, W, 3)[0].to(gpu)
    kp = torch.zeros((0, 2), dtype=torch.float32, device=gpu)
    out = ops.match_cov(depth, kp, torch.zeros((0, 3), device=gpu), None, 80.0, 80.0, 48.0, 32.0)
    assert out.shape == (0, 3, 3)
    valid, count = ops.obs_filter(torch.zeros(0, dtype=torch.bool, device=gpu), torch.zeros((0, 3, 3), dtype=torch.float64, device=gpu),
                                  torch.zeros((0, 3, 3), dtype=torch.float64, device=gpu), None)
    assert valid.numel() == 0 and int(count.item()) == 0


def test_pgo_lost_track_and_masked_rows(gpu):
    """< min_points valid observations: no optimisation, pose = prior, steps = 0 (MACVO.py:303-307); masked rows are
    ignored exactly as if they had been dropped from the problem."""
    from macvo_amd import ops
    from oracle import pgo, se3
    from tests.test_gpu_backend import _to_batch

    prob, _ = pgo.make_synthetic_problem(n=60, seed=21)
    batch = _to_batch([prob], gpu)
    valid = torch.zeros(60, dtype=torch.bool)
    valid[:7] = True
    batch.valid = valid.to(gpu)
    pose, info = ops.pgo_solve(batch, "disp", min_points=10)
    assert torch.equal(pose[0].cpu(), prob.init_pose.double()) and int(info[0, 1].item()) == 0
    # masked solve == solve of the compacted problem
    keep = torch.rand(60, generator=torch.Generator().manual_seed(1)) > 0.3
    batch.valid = keep.to(gpu)
    pose_m, info_m = ops.pgo_solve(batch, "disp", min_points=10)
    sub = pgo.PGOProblem(prob.init_pose, prob.K, prob.baseline, prob.pos_Tw[keep], prob.cov_Tw[keep], prob.pixel2_uv[keep],
                         prob.pixel2_d[keep], prob.pixel2_disp[keep], prob.pixel2_disp_cov[keep], prob.pixel2_uv_cov[keep],
                         prob.obs2_covTc[keep])
    ref = pgo.solve(sub, "disp")
    dt, dr = se3.pose_error(ref.pose, pose_m[0].cpu())
    assert dt < 1e-8 and dr < 1e-8 and int(info_m[0, 1].item()) == ref.steps


def test_pgo_more_points_than_threads_and_outliers(gpu):
    """N > 256 takes the generic (re-reading) path of the 4-wave kernel; heavy outliers exercise the Huber branch."""
    from macvo_amd import ops
    from oracle import pgo, se3
    from tests.test_gpu_backend import _to_batch

    for graph in ("disp", "icp"):
        prob, _ = pgo.make_synthetic_problem(n=700, seed=33, outlier_frac=0.4)
        pose, info = ops.pgo_solve(_to_batch([prob], gpu), graph)
        ref = pgo.solve(prob, graph)
        dt, dr = se3.pose_error(ref.pose, pose[0].cpu())
        assert dt < 1e-7 and dr < 1e-7 and int(info[0, 1].item()) == ref.steps


def test_pgo_batched_throughput_variant_matches(gpu):
    """nprob >= 512 dispatches the 1-wave-per-problem kernel: same answers as the 4-wave one."""
    from macvo_amd import ops
    from oracle import pgo
    from tests.test_gpu_backend import _to_batch

    base = [pgo.make_synthetic_problem(n=50 + 13 * k, seed=40 + k)[0] for k in range(6)]
    small, _ = ops.pgo_solve(_to_batch(base, gpu), "disp")
    big, _ = ops.pgo_solve(_to_batch([base[k % 6] for k in range(600)], gpu), "disp")
    for k in range(600):
        assert (big[k] - small[k % 6]).abs().max().item() < 1e-9


def test_volume_and_lookup_degenerate_sizes(gpu):
    from macvo_amd import ops
    from oracle import corr

    g = torch.Generator().manual_seed(2)
    f1, f2 = torch.randn(1, 16, 2, 3, generator=g), torch.randn(1, 16, 2, 3, generator=g)
    vol = ops.corr_volume(f1.to(gpu), f2.to(gpu))
    assert (vol.cpu().double() - corr.corr_volume(f1, f2, torch.float64)).abs().max() < 1e-5
    coords = corr.coords_grid(1, 2, 3) + 0.3
    out = ops.corr_lookup(vol, coords.to(gpu), 4)
    torch.testing.assert_close(out.cpu(), corr.corr_lookup(vol.cpu(), coords, 4), rtol=1e-5, atol=1e-5)
    # different source / target resolutions (H1 x W1 queries into H2 x W2 slices)
    f1, f2 = torch.randn(2, 32, 5, 7, generator=g), torch.randn(2, 32, 9, 4, generator=g)
    vol = ops.corr_volume(f1.to(gpu), f2.to(gpu))
    assert vol.shape == (2 * 35, 1, 9, 4)
    assert (vol.cpu().double() - corr.corr_volume(f1, f2, torch.float64)).abs().max() < 1e-4
    coords = torch.rand(2, 2, 5, 7, generator=g) * 8
    torch.testing.assert_close(ops.corr_lookup(vol, coords.to(gpu), 4).cpu(), corr.corr_lookup(vol.cpu(), coords, 4), rtol=1e-5, atol=1e-4)


def test_hot_path_survives_frame_without_keypoints(gpu):
    from macvo_amd.pipeline import Camera, FrameInputs, HotPath, HotPathConfig

    cam, frames, _ = synth.make_sequence(3, 128, 160, C=32, iters=1, seed=4)
    frames[1]["logcov"] = torch.full_like(frames[1]["logcov"], float("nan"))   # selector finds nothing on frame 1
    hot = HotPath(Camera(**cam), HotPathConfig(kp_mask_width=16, edgewidth=16), gpu)
    ins = [FrameInputs(**{k: v.to(gpu) for k, v in f.items()}) for f in frames]
    torch.cuda.synchronize()
    hot.initialize(ins[0])
    p0 = hot.pose.clone()
    r1 = hot.step(ins[1])
    assert r1.kp0_uv.shape[0] == 0 and torch.equal(hot.pose, p0)
    r2 = hot.step(ins[2])       # depth of frame 1 is NaN-free (only its covariances were NaN) -> tracking resumes
    assert r2.kp0_uv.shape[0] > 0 and torch.isfinite(hot.pose).all()
