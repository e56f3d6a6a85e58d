"""Ground truth shipped with the reference (its unit-test TartanAir asset: GT depth, GT flow, GT poses) through the
hot path: with perfect frontend outputs the covariance-weighted two-frame PGO must reproduce the ground-truth motion.
This pins the oracle's geometry (NED axes, pixel2point, SE3 composition, residual definitions, Jacobians) to data the
reference's own tests use — independent of every synthetic generator in this repository."""
import pytest
import torch

from tests import synth


def _pose_err(a, b):
    from oracle import se3

    return se3.pose_error(a.double(), b.double())


@pytest.mark.parametrize("graph", ["disp", "reproj", "icp"])
def test_oracle_recovers_ground_truth_motion(graph):
    from oracle.pipeline import OracleHotPath

    cam, frames, poses = synth.tartanair_sequence()
    ora = OracleHotPath(cam, dict(graph_type=graph))
    ora.initialize(frames[0], init_pose=poses[0])
    for t in range(1, len(frames)):
        torch.manual_seed(50 + t)
        r = ora.step(frames[t])
        assert r["n_valid"] >= 150, r["n_valid"]
        dt, dr = _pose_err(poses[t], r["pose"])
        # inter-frame motion is 0.06-0.16 m / 0.013-0.024 rad; GT flow is quantised to 1/64 px and observation depths are
        # read at truncated pixel positions (Frontend.py:117): measured 0.1-0.4 mm / 3e-5-8e-5 rad for the image-space
        # graphs.  The ICP graph compares 3-D points built from those truncated-pixel depths and stops at the reference's
        # 10-step cap before converging (1-17 mm): it only has to remove most of the prior's error.
        d0, r0 = _pose_err(poses[t], poses[t - 1])
        if graph == "icp":
            assert dt < 0.2 * d0 and dr < 0.5 * r0, (graph, t, dt, dr)
        else:
            assert dt < 1e-3 and dr < 2e-4, (graph, t, dt, dr)
            assert dt < 0.01 * d0 and dr < 0.01 * r0, "the solve must remove at least 99 % of the prior's error"
        ora.pose = poses[t].float()


@pytest.mark.gpu
@pytest.mark.parametrize("driver", ["native", "python"])
def test_hip_path_on_reference_ground_truth(gpu, driver):
    from macvo_amd.pipeline import Camera, FrameInputs, HotPath, HotPathConfig, NativeHotPath
    from oracle.pipeline import OracleHotPath

    cam, frames, poses = synth.tartanair_sequence()
    ins = [FrameInputs(**{k: v.to(gpu) for k, v in fr.items()}) for fr in frames]
    torch.cuda.synchronize()
    ora = OracleHotPath(cam, {})
    hot = (NativeHotPath if driver == "native" else HotPath)(Camera(**cam), HotPathConfig(), gpu)
    ora.initialize(frames[0], init_pose=poses[0])
    hot.initialize(ins[0], init_pose=poses[0].float())
    for t in range(1, len(frames)):
        torch.manual_seed(50 + t)
        ro = ora.step(frames[t])
        torch.manual_seed(50 + t)
        rh = hot.step(ins[t])
        torch.cuda.synchronize()
        assert torch.equal(rh.kp0_uv.cpu(), ro["kp0_uv"]), t                  # bit-exact keypoints
        assert int(rh.n_valid.item()) == ro["n_valid"]
        dt, dr = _pose_err(ro["pose"], rh.pose.cpu())
        assert dt <= 1e-4 and dr <= 1e-4, (t, dt, dr)                           # north_star parity bar
        gt, gr = _pose_err(poses[t], rh.pose.cpu())
        assert gt < 1e-3 and gr < 2e-4, (t, gt, gr)                             # and the truth itself
        ora.pose = poses[t].float()
        hot.pose = poses[t].float().to(gpu)
