"""The PGO kernel's arithmetic and loop structure, replayed on the host (tests/c_abi/pgo_twin.cpp includes the kernel's own
mac-vo_amd/csrc/pgo_math.h), against the oracle and against the reference golden — CPU, no GPU needed.  The GPU suite then holds
the kernel to this twin (test_gpu_backend.py::test_pgo_kernel_equals_host_twin)."""
import os

import numpy as np
import pytest
import torch

from tests import pgo_twin
from tests.test_gpu_backend import _to_batch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CPU = torch.device("cpu")

CASES = [dict(n=200, seed=6), dict(n=200, seed=7, outlier_frac=0.1), dict(n=37, seed=8), dict(n=12, seed=9),
         dict(n=500, seed=10, trans_sigma=0.4, rot_sigma=0.08), dict(n=64, seed=11, outlier_frac=0.3)]


@pytest.mark.parametrize("nw", [4, 1])
@pytest.mark.parametrize("graph", ["disp", "reproj", "icp"])
def test_twin_matches_oracle(graph, nw):
    """Same bar as the kernel's own test (test_pgo_matches_oracle): pose at fp64 roundoff, the same number of LM steps, the loss.
    n = 500 exercises the several-points-per-thread (55-value) build, the rest the one-point-per-thread (28-value) one; nw = 1 is
    the throughput variant (64 threads per problem: every case with n > 64 takes the 55-value path there)."""
    from oracle import pgo, se3

    probs = [pgo.make_synthetic_problem(**c)[0] for c in CASES]
    pose, info = pgo_twin.solve(_to_batch(probs, CPU), graph, nw=nw)
    for k, p in enumerate(probs):
        ref = pgo.solve(p, graph)
        dt, dr = se3.pose_error(ref.pose, pose[k])
        assert dt <= 1e-8 and dr <= 1e-8, (k, dt, dr)
        assert int(info[k, 1]) == ref.steps, (k, info[k], ref.steps)
        assert int(info[k, 2]) == ref.reject_count, (k, info[k], ref.reject_count)
        assert info[k, 0].item() == pytest.approx(ref.loss, rel=1e-8, abs=1e-12)


@pytest.mark.parametrize("graph", ["icp", "reproj", "disp"])
@pytest.mark.parametrize("variant", ["", "_r16"])
def test_twin_vs_reference_golden(graph, variant):
    """The twin against what the reference's in-tree optimizer code produced on the PyPose shim (tests/golden/pgo.npz): pose, outer
    LM steps, reject count of the last step, final loss — the asserts of test_gpu_golden.py::test_pgo_vs_reference_golden."""
    from macvo_amd import ops
    from oracle import pgo, se3

    z = np.load(os.path.join(GOLD, "pgo.npz"))
    probs, refs, stats = [], [], []
    ci = 0
    while f"{graph}_{ci}_pose{variant}" in z:
        n, seed, of, ts, rs = [float(v) for v in z[f"{graph}_{ci}_case"]]
        probs.append(pgo.make_synthetic_problem(n=int(n), seed=int(seed), outlier_frac=of, trans_sigma=ts, rot_sigma=rs)[0])
        refs.append(torch.from_numpy(z[f"{graph}_{ci}_pose{variant}"]))
        stats.append(z[f"{graph}_{ci}_stats{variant}"])
        ci += 1
    assert ci >= 7
    lm = ops.lm_default_params()
    lm.stop_on_reject = 1 if variant == "" else 16
    for spec in (1, 0):
        pose, info = pgo_twin.solve(_to_batch(probs, CPU), graph, lm, spec=spec)
        for k, ref in enumerate(refs):
            dt, dr = se3.pose_error(ref, pose[k])
            assert dt <= 1e-8 and dr <= 1e-8, (k, dt, dr)
            steps, rej, loss, _ = [float(v) for v in stats[k]]
            assert int(info[k, 1]) == int(steps) and int(info[k, 2]) == int(rej), (spec, k, info[k].tolist(), stats[k].tolist())
            assert abs(float(info[k, 0]) - loss) <= 1e-6 * max(1.0, abs(loss)), (k, float(info[k, 0]), loss)


@pytest.mark.parametrize("graph", ["disp", "reproj", "icp"])
def test_twin_speculative_rounds_keep_the_bits(graph):
    """Speculative reject rounds (four trials at once) against one trial at a time: identical bits, and the rounds really run
    (spec = 2 reports rounds * 1000 + loop iterations)."""
    from macvo_amd import ops
    from oracle import pgo

    probs = [pgo.make_synthetic_problem(n=200, seed=s)[0] for s in (6, 1, 2, 3)]
    probs += [pgo.make_synthetic_problem(n=150, seed=21, outlier_frac=0.2, trans_sigma=0.3, rot_sigma=0.05)[0]]
    for sor in (1, 16):
        lm = ops.lm_default_params()
        lm.stop_on_reject = sor
        b = _to_batch(probs, CPU)
        p1, i1 = pgo_twin.solve(b, graph, lm, spec=1)
        p0, i0 = pgo_twin.solve(b, graph, lm, spec=0)
        assert torch.equal(p1, p0) and torch.equal(i1, i0)
        _, i2 = pgo_twin.solve(b, graph, lm, spec=2)
        assert torch.equal(i2[:, :3], i1[:, :3])
        if graph == "disp" and sor == 1:
            assert (i2[:, 3] >= 1000).any(), i2


def test_twin_valid_mask_and_min_points():
    """Rows masked out through `valid` do not contribute (== the problem without them); fewer than min_points valid rows => the
    prior pose comes back with steps = 0 (MACVO.py:303-307)."""
    from oracle import pgo

    prob = pgo.make_synthetic_problem(n=120, seed=4)[0]
    b = _to_batch([prob], CPU)
    keep = torch.ones(120, dtype=torch.bool)
    keep[::3] = False
    b.valid = keep
    pose_m, info_m = pgo_twin.solve(b, "disp")
    sub = pgo.make_synthetic_problem(n=120, seed=4)[0]
    for f in ("pos_Tw", "pixel2_uv", "cov_Tw", "pixel2_d", "pixel2_disp", "pixel2_disp_cov", "pixel2_uv_cov", "obs2_covTc"):
        setattr(sub, f, getattr(sub, f)[keep])
    ref = pgo.solve(sub, "disp")
    from oracle import se3
    dt, dr = se3.pose_error(ref.pose, pose_m[0])
    assert dt <= 1e-8 and dr <= 1e-8 and int(info_m[0, 1]) == ref.steps
    pose_l, info_l = pgo_twin.solve(b, "disp", min_points=100)
    assert int(info_l[0, 1]) == 0
    assert torch.equal(pose_l[0].float(), prob.init_pose.float())


@pytest.mark.parametrize("graph", ["disp", "reproj", "icp"])
def test_twin_random_sweep_vs_oracle(graph):
    """24 random problems per graph (12 .. 256 points, up to 30 % outliers, initial errors up to 0.5 m / 0.1 rad): the same pose (1e-8), the
    same number of LM steps and of rejections as the oracle on every one — the accept / reject decisions of the explicit-FMA, lean-build
    arithmetic do not hang on roundoff (a 180-problem sweep of the same generator, run while developing it, had no mismatch either)."""
    import random

    from oracle import pgo, se3

    rnd = random.Random(5)
    cfgs = [dict(n=rnd.choice([12, 37, 64, 100, 200, 256]), seed=2000 + k, outlier_frac=rnd.choice([0, 0, 0.1, 0.3]),
                 trans_sigma=rnd.choice([0.1, 0.3, 0.5]), rot_sigma=rnd.choice([0.02, 0.06, 0.1])) for k in range(24)]
    probs = [pgo.make_synthetic_problem(**c)[0] for c in cfgs]
    pose, info = pgo_twin.solve(_to_batch(probs, CPU), graph)
    for k, p in enumerate(probs):
        ref = pgo.solve(p, graph)
        dt, dr = se3.pose_error(ref.pose, pose[k])
        assert dt <= 1e-8 and dr <= 1e-8, (cfgs[k], dt, dr)
        assert (int(info[k, 1]), int(info[k, 2])) == (ref.steps, ref.reject_count), (cfgs[k], info[k].tolist(), ref.steps, ref.reject_count)
