"""GPU: HIP path vs golden vectors produced by the REAL reference modules (tests/golden/*.npz)."""
import pytest
import torch

from tests.test_oracle_golden import _sel_inputs, load

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["small", "full"])
def test_selectors_vs_reference_golden(gpu, name):
    from macvo_amd import ops

    z = load("selector")
    H, W, mw, fc, d0, d0c, d1, d1c = _sel_inputs(z, name)
    dv = lambda t: t.to(gpu)  # noqa: E731
    torch.manual_seed(1234)
    px = ops.kp_select("nodepth", H, W, flow_cov=dv(fc), kernel_size=7, mask_width=mw, max_match_cov=100.0).finish(200)
    assert torch.equal(px.cpu(), z[f"{name}_nodepth_px"])
    torch.manual_seed(4321)
    px = ops.kp_select("full", H, W, flow_cov=dv(fc), depth0=dv(d0), depth0_cov=dv(d0c), depth1=dv(d1), depth1_cov=dv(d1c),
                       kernel_size=7, mask_width=mw, max_depth=320.0 * 0.25, max_depth_cov=250.0, max_match_cov=100.0).finish(200)
    assert torch.equal(px.cpu(), z[f"{name}_full_px"])
    torch.manual_seed(7)
    px = ops.kp_select("mapping", H, W, depth0=dv(d0), depth0_cov=dv(d0c), mask_width=mw, max_depth=20.0, max_depth_cov=0.2).finish(300)
    assert torch.equal(px.cpu(), z[f"{name}_mapping_px"])


def test_covariance_vs_reference_golden(gpu):
    from macvo_amd import ops

    z = load("covariance")
    K = [float(v) for v in z["K"]]
    depth = z["depth"].to(gpu)
    fc = z["flow_cov_in"].clone().to(gpu)
    out = ops.match_cov(depth, z["kp_int"].to(gpu), fc, None, *K)
    assert torch.equal(fc.cpu(), z["flow_cov_after"])                         # in-place clamp, bit-exact
    # tolerances = 5 x the measured bound (profiles/probes/cov_tolerance.py: max relative error 2.6e-6 / 1.0e-5 / 4.5e-7 / 2.4e-7):
    # the kernel replays the reference's fp32 op order; what is left is the summation order of the 961-tap reductions
    torch.testing.assert_close(out.cpu(), z["cov_int_flowcov"], rtol=5e-5, atol=1e-7)
    out = ops.match_cov(depth, z["kp_float"].to(gpu), z["flow_cov_in"].clone().to(gpu), None, *K)
    torch.testing.assert_close(out.cpu(), z["cov_float_flowcov"], rtol=5e-5, atol=1e-7)
    s0 = torch.ones(z["kp_int"].shape[0], 3) * 0.25
    s0[:, 2] = 0
    out = ops.match_cov(depth, z["kp_int"].to(gpu), s0.to(gpu), z["depth_cov_kp"].to(gpu), *K, use_patch_var=True)
    torch.testing.assert_close(out.cpu(), z["cov_int_default_sigma"], rtol=5e-6, atol=1e-8)
    out = ops.match_cov(depth, z["kp_int"].to(gpu), s0.to(gpu), z["depth_cov_kp"].to(gpu), *K, use_patch_var=False)
    torch.testing.assert_close(out.cpu(), z["cov_int_nodefault"], rtol=2e-6, atol=1e-8)


@pytest.mark.parametrize("graph", ["icp", "reproj", "disp"])
@pytest.mark.parametrize("variant", ["", "_r16"])
def test_pgo_vs_reference_golden(gpu, graph, variant):
    """mv_pgo_solve vs the reference's in-tree optimizer code run on the PyPose shim (pgo.npz): pose, number of outer LM
    steps, reject_count of the last step and final loss, for both readings of StopOnPlateau's reject rule
    (``mvLMParams.stop_on_reject`` = 1, the default, and = 16).  Cases 4-6 enter the reject loop mid-solve."""
    from macvo_amd import ops
    from oracle import pgo, se3
    from tests.test_gpu_backend import _to_batch

    z = load("pgo")
    probs, refs, stats = [], [], []
    ci = 0
    while f"{graph}_{ci}_pose{variant}" in z:
        n, seed, of, ts, rs = [float(v) for v in z[f"{graph}_{ci}_case"]]
        probs.append(pgo.make_synthetic_problem(n=int(n), seed=int(seed), outlier_frac=of, trans_sigma=ts, rot_sigma=rs)[0])
        refs.append(z[f"{graph}_{ci}_pose{variant}"])
        stats.append(z[f"{graph}_{ci}_stats{variant}"])
        ci += 1
    assert ci >= 7
    lm = ops.lm_default_params()
    assert lm.stop_on_reject == 1
    lm.stop_on_reject = 1 if variant == "" else 16
    pose, info = ops.pgo_solve(_to_batch(probs, gpu), graph, lm)
    info = info.cpu()
    for k, ref in enumerate(refs):
        dt, dr = se3.pose_error(ref, pose[k].cpu())
        assert dt <= 1e-4 and dr <= 1e-4, (k, dt, dr)      # north_star tolerance
        assert dt <= 1e-8 and dr <= 1e-8, (k, dt, dr)      # what is actually achieved
        steps, rej, loss, _ = [float(v) for v in stats[k]]
        assert int(info[k, 1]) == int(steps) and int(info[k, 2]) == int(rej), (k, info[k].tolist(), stats[k].tolist())
        # the robust loss has a slope of ~1e2 per metre of pose change: poses equal to 1e-8 leave it equal to ~1e-6
        assert abs(float(info[k, 0]) - loss) <= 1e-6 * max(1.0, abs(loss)), (k, float(info[k, 0]), loss)


def test_obs_filter_vs_reference_classes(gpu):
    """mv_obs_filter (flags 1 / 2 / 4 and their combinations) vs the masks the REAL filter classes produced
    (tests/golden/filters.npz): NaN / +-inf covariances, depths exactly on the gates, the 2-sigma test at equality, a NaN
    variance, and the -1 "no covariance" placeholder that switches LikelyFrontOfCamFilter off for every row."""
    import os

    import numpy as np

    from macvo_amd import ops

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "filters.npz"))
    T = lambda k: torch.from_numpy(z[k])  # noqa: E731
    n = T("d1").shape[0]
    for tag in ("", "_placeholder"):
        c1 = T("c1").clone()
        if tag:
            c1[40] = -1.0
        vals = torch.zeros(11, n)
        vals[0], vals[3], vals[4], vals[7] = T("d1")[:, 0], c1[:, 0], T("d2")[:, 0], T("c2")[:, 0]
        masks = {1: T("sanity" + tag), 2: T("depth" + tag), 4: T("front" + tag)}
        for flags in (1, 2, 4, 3, 5, 6, 7):
            want = torch.ones(n, dtype=torch.bool)
            for bit, m in masks.items():
                if flags & bit:
                    want &= m
            valid, count = ops.obs_filter(None, T("cov1").to(gpu), T("cov2").to(gpu), vals.to(gpu), flags, 0.05, 10.0)
            assert torch.equal(valid.cpu(), want), (tag, flags)
            assert int(count.item()) == int(want.sum())
        inb = torch.arange(n) % 3 != 0
        valid, _ = ops.obs_filter(inb.to(gpu), T("cov1").to(gpu), T("cov2").to(gpu), vals.to(gpu), 7, 0.05, 10.0)
        assert torch.equal(valid.cpu(), inb & masks[1] & masks[2] & masks[4])
