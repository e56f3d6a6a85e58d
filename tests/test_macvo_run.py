"""The reference's own, unmodified ``Odometry/MACVO.py`` loop (``from_config`` -> ``receive_frames`` -> ``run_pair`` per frame ->
``terminate``) as the pin of the whole path:

  * ``tests/golden/macvo_run.npz`` = what that loop, with every module the REFERENCE's class on the CPU, writes to ``tensor_map.npz`` /
    ``poses.npy`` on the reference's unit-test TartanAir asset (``tests/refrun.py --golden``; four configurations);
  * CPU: the golden regenerates (when the reference tree is present), and ``oracle/pipeline.py`` — the restatement every other parity
    test leans on — reproduces the reference run's keypoints bit for bit and its poses to 1e-6;
  * GPU: the SAME loop with only the ``type:`` strings swapped to the ``HIP_*`` plugins reproduces the golden (keypoints and every
    gathered store bit-exact, covariances 5e-5, poses 1e-4 m / 1e-4 rad) — and, run live side by side on the GPU box's host, the
    640x480 synthetic stream of the benchmark.

The loops run in fresh interpreters (``tests/refrun.py``): the reference's real ABCs and this repository's mirror classes must not mix."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import refrun

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "macvo_run.npz")
needs_ref = pytest.mark.skipif(refrun.reference_root() is None,
                               reason="needs the reference's Python tree (/root/reference or oracle/_ref/pyref built by oracle/build_ref.py)")


def _golden(case):
    z = np.load(GOLDEN)
    return {k[len(case) + 1:]: z[k] for k in z.files if k.startswith(case + "/")}


def _run(tmp_path, case, mode, frames=0):
    out = tmp_path / f"{case}_{mode}.npz"
    cmd = [sys.executable, os.path.join(ROOT, "tests", "refrun.py"), "--mode", mode, "--case", case, "--out", str(out)]
    if frames:
        cmd += ["--frames", str(frames)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-6000:]
    info = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    return dict(np.load(out)), info


@needs_ref
@pytest.mark.parametrize("case", refrun.GOLDEN_CASES)
def test_reference_loop_regenerates_the_golden(tmp_path, case):
    run, info = _run(tmp_path, case, "ref")
    assert info["classes"]["Optimizer"] == "TwoFrame_PGO" and info["classes"]["ObsCovModel"] == "MatchCovariance"
    # same reference code, same exact-arithmetic inputs; fp64 LAPACK / libm may differ in the last bits between hosts
    assert refrun.compare_runs(_golden(case), run, pose_tol=1e-6) == []


def _relative_motion_errors(est_poses, gt_poses):
    from oracle import se3

    est, gt = torch.from_numpy(est_poses).double(), torch.from_numpy(gt_poses).double()
    errs = []
    for t in range(1, len(est)):
        d_est = se3.se3_mul(se3.se3_inv(est[t - 1]), est[t])
        d_gt = se3.se3_mul(se3.se3_inv(gt[t - 1]), gt[t])
        errs.append(se3.pose_error(d_est, d_gt))
    return errs


@pytest.mark.parametrize("case", ["tartan_fast", "tartan_reproj"])
def test_golden_run_recovers_the_ground_truth_motion(case):
    g = _golden(case)
    assert g["map/frames//pose"].shape == (4, 7) and not g["map/frames//need_interp"].any()
    for dt, dr in _relative_motion_errors(g["map/frames//pose"], g["gt_poses"]):
        assert dt < 2e-3 and dr < 5e-4, (dt, dr)          # inter-frame motion 0.06-0.16 m / 0.013-0.024 rad


@pytest.mark.parametrize("case", ["tartan_fast", "tartan_reproj"])
def test_oracle_pipeline_reproduces_the_reference_run(case):
    """oracle/pipeline.OracleHotPath (a re-enactment of run_pair) against what the REAL run_pair produced."""
    from oracle.pipeline import OracleHotPath

    g = _golden(case)
    spec = refrun.CASES[case]
    cam, maps, _ = refrun.tartanair_maps()
    ora = OracleHotPath(cam, dict(graph_type=spec["graph"], selector="nodepth" if spec["selector"].endswith("NoDepth") else "full",
                                  mapping=spec["mapping"]))
    torch.manual_seed(1234)
    ora.initialize(dict(fmap1=torch.zeros(2, 8, 2, 2), fmap2=torch.zeros(2, 8, 2, 2), coords=torch.zeros(0, 2, 2, 2, 2),
                        flow=maps[0]["flow"], cov_exp=maps[0]["cov"]))
    ranges = g["map/edge/frame2match/ranges"]
    for t in range(1, len(maps)):
        r = ora.step(dict(fmap1=torch.zeros(2, 8, 2, 2), fmap2=torch.zeros(2, 8, 2, 2), coords=torch.zeros(0, 2, 2, 2, 2),
                          flow=maps[t]["flow"], cov_exp=maps[t]["cov"]))
        lo, n = int(ranges[t, 0, 0]), int(ranges[t, 0, 1])                         # the frame's rows of the match store
        kp_ref = g["map/match//pixel1_uv"][lo:lo + n]
        kp_ora = r["tracked"]["kp0_uv"][r["mask"]].float().numpy()
        assert np.array_equal(kp_ref, kp_ora), (case, t)                          # bit-exact keypoints, after the reference's own filters
        assert np.array_equal(g["map/match//pixel2_uv"][lo:lo + n], r["tracked"]["kp1_uv"][r["mask"]].numpy())
        c1 = r["cov1"][r["mask"]].numpy()
        assert np.allclose(g["map/match//obs2_covTc"][lo:lo + n], c1, rtol=1e-6, atol=1e-12)
        assert np.abs(g["map/frames//pose"][t] - r["pose"].numpy()).max() < 1e-6, (case, t)


def test_golden_cases_cover_both_selectors_all_graphs_and_both_frontend_wirings():
    z = np.load(GOLDEN)
    cls = {c: json.loads(str(z[f"{c}/classes"])) for c in refrun.GOLDEN_CASES}
    assert {v["KeypointSelector"] for v in cls.values()} == {"CovAwareSelector", "CovAwareSelector_NoDepth"}
    assert {v["Frontend"] for v in cls.values()} == {"Replay_FlowFormerCovFrontend", "FrontendCompose"}
    assert {refrun.CASES[c]["graph"] for c in refrun.GOLDEN_CASES} == {"icp", "reproj", "disp"}
    for c in refrun.GOLDEN_CASES:
        assert z[f"{c}/map/match//pixel1_uv"].shape[0] > 450                     # ~190 tracked keypoints per frame


HIP_CLASSES = {"KeypointSelector": "HIP_", "MappointSelector": "HIP_MappingPointSelector", "ObsCovModel": "HIP_MatchCovariance",
               "Optimizer": "HIP_TwoFrame_PGO"}


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("case", refrun.GOLDEN_CASES)
def test_hip_plugins_inside_the_reference_loop_match_the_golden(gpu, tmp_path, case):
    run, info = _run(tmp_path, case, "hip")
    for k, v in HIP_CLASSES.items():
        assert info["classes"][k].startswith(v), info["classes"]
    if refrun.CASES[case]["frontend"] == "replay":
        assert info["classes"]["Frontend"] == "HIP_FlowFormerCovFrontend"
    assert refrun.compare_runs(_golden(case), run, pose_tol=1e-4) == []


@pytest.mark.gpu
@needs_ref
def test_hip_plugins_inside_the_reference_loop_on_the_benchmark_stream(gpu, tmp_path):
    """640x480 synthetic stream (the bench workload's flow / covariance maps), reference classes on this host's CPU vs HIP plugins, live."""
    ref, _ = _run(tmp_path, "synth_fast", "ref", frames=6)
    hip, info = _run(tmp_path, "synth_fast", "hip", frames=6)
    assert info["classes"]["Optimizer"] == "HIP_TwoFrame_PGO"
    assert ref["map/match//pixel1_uv"].shape[0] > 800
    assert refrun.compare_runs(ref, hip, pose_tol=1e-4) == []
