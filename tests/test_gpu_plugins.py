"""GPU: the drop-in plugin classes, called exactly the way Odometry/MACVO.py calls the reference's (run_pair :197-311)."""
from types import SimpleNamespace

import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu


class _Bundle:
    def __init__(self, data, index=None):
        self.data, self.index = data, index


class _FakeMap:
    """The four accessors TwoFrame_PGO.get_graph_data / write_graph_data use (Module/Map/VisualMap.py)."""

    def __init__(self, prob):
        n = prob.pos_Tw.shape[0]
        self.frames = SimpleNamespace(data={"pose": torch.zeros(2, 7)})
        self.frames.data["pose"][:, 6] = 1
        self._frame = _Bundle({"K": prob.K[None], "pose": prob.init_pose[None].clone(), "baseline": torch.tensor([prob.baseline])})
        self._obs = _Bundle({"pixel2_uv": prob.pixel2_uv, "pixel2_d": prob.pixel2_d, "pixel2_disp": prob.pixel2_disp,
                             "pixel2_disp_cov": prob.pixel2_disp_cov, "pixel2_uv_cov": prob.pixel2_uv_cov, "obs2_covTc": prob.obs2_covTc})
        self._pts = _Bundle({"pos_Tw": prob.pos_Tw, "cov_Tw": prob.cov_Tw})
        self.frames.__dict__["__getitem__"] = None
        self.n = n

    def get_frame2match(self, frame):
        return self._obs

    def get_match2point(self, obs):
        return self._pts


class _Frames(dict):
    pass


@pytest.mark.parametrize("parallel", [False, True])
@pytest.mark.parametrize("graph", ["disp", "icp"])
def test_hip_two_frame_pgo_plugin_protocol(gpu, parallel, graph):
    from macvo_amd.plugins import HIP_TwoFrame_PGO
    from oracle import pgo, se3

    prob, _ = pgo.make_synthetic_problem(n=150, seed=12)
    cfg = SimpleNamespace(device="cpu", vectorize=True, parallel=parallel, graph_type=graph, autodiff=False)
    HIP_TwoFrame_PGO.is_valid_config(cfg)
    opt = HIP_TwoFrame_PGO(cfg)
    fmap = _FakeMap(prob)

    class Frames:
        data = fmap.frames.data

        def __getitem__(self, idx):
            return fmap._frame

    fmap.frames = Frames()
    opt.write_map(fmap)                                   # no job yet: must be a no-op
    frame_idx = torch.tensor([1])
    opt.start_optimize(opt.get_graph_data(fmap, frame_idx))
    opt.write_map(fmap)                                   # joins the job, writes float32 pose
    ref = pgo.solve(prob, graph)
    got = fmap.frames.data["pose"][1]
    assert got.dtype == torch.float32
    assert torch.equal(got, ref.pose.float()) or se3.pose_error(ref.pose, got.double()) < (1e-6, 1e-6)
    opt.terminate()


def test_hip_match_covariance_plugin_contract(gpu):
    from macvo_amd.plugins import HIP_MatchCovariance
    from oracle import covariance

    H, W, n = 240, 320, 64
    depth, dcov = synth.depth_maps(H, W, 3)
    kp = synth.keypoints(n, H, W, 6).to(gpu)
    frame = SimpleNamespace(fx=160.0, fy=160.0, cx=160.0, cy=120.0)
    cfg = SimpleNamespace(device="cuda", kernel_size=31, match_cov_default=0.25, min_depth_cov=0.05, min_flow_cov=0.25)
    HIP_MatchCovariance.is_valid_config(cfg)
    model = HIP_MatchCovariance(cfg)
    dest = SimpleNamespace(depth=depth.to(gpu), cov=dcov.to(gpu))
    fc = torch.rand(n, 3) * 0.3
    fc[:, 2] = 0
    fc_dev = fc.to(gpu)
    out = model.estimate(frame, kp, dest, None, fc_dev)
    assert out.device.type == "cpu" and out.dtype == torch.float64 and out.shape == (n, 3, 3)
    fc_ref = fc.clone()
    ref = covariance.match_covariance(kp.cpu(), depth, None, fc_ref, 160.0, 160.0, 160.0, 120.0)
    torch.testing.assert_close(out, ref, rtol=2e-3, atol=1e-7)
    assert torch.equal(fc_dev.cpu(), fc_ref)             # caller's tensor was clamped in place
    # flow_cov None -> default sigma, given depth variance branch
    dc = dcov[0, 0, kp[:, 1].cpu(), kp[:, 0].cpu()].contiguous()
    out2 = model.estimate(frame, kp, dest, dc.to(gpu), None)
    torch.testing.assert_close(out2, covariance.match_covariance(kp.cpu(), depth, dc, None, 160.0, 160.0, 160.0, 120.0), rtol=2e-4, atol=1e-8)


def test_hip_selector_plugins(gpu):
    from macvo_amd.plugins import HIP_CovAwareSelector, HIP_CovAwareSelector_NoDepth, HIP_MappingPointSelector
    from oracle import selector

    H, W = 240, 320
    fc = synth.flow_cov_maps(H, W, 2)
    d0, d0c = synth.depth_maps(H, W, 3)
    d1, d1c = synth.depth_maps(H, W, 4)
    frame = SimpleNamespace(fx=320.0, frame_baseline=0.25, height=H, width=W)
    dep0 = SimpleNamespace(depth=d0.to(gpu), cov=d0c.to(gpu), mask=None)
    dep1 = SimpleNamespace(depth=d1.to(gpu), cov=d1c.to(gpu), mask=None)
    match = SimpleNamespace(cov=fc.to(gpu), mask=None)

    s = HIP_CovAwareSelector_NoDepth(SimpleNamespace(device="cuda", kernel_size=7, mask_width=32, max_match_cov=100.0))
    torch.manual_seed(5)
    px = s.select_point(frame, 200, dep0, dep1, match)
    torch.manual_seed(5)
    assert torch.equal(px.cpu(), selector.cov_aware_selector_nodepth(fc.clone(), 200, 7, 32, 100.0)[0])
    grid = s.select_point(frame, 200, dep0, dep1, None)   # GridSelector fallback
    assert grid.shape[1] == 2 and int(grid.min()) >= 32

    cfg = SimpleNamespace(device="cuda", kernel_size=7, mask_width=32, max_depth="auto", max_depth_cov=250.0, max_match_cov=100.0)
    s2 = HIP_CovAwareSelector(cfg)
    torch.manual_seed(6)
    px = s2.select_point(frame, 200, dep0, dep1, match)
    assert cfg.max_depth == 80.0
    torch.manual_seed(6)
    assert torch.equal(px.cpu(), selector.cov_aware_selector(d0, d0c, d1, d1c, fc.clone(), 200, 80.0, 7, 32, 250.0, 100.0)[0])

    s3 = HIP_MappingPointSelector(SimpleNamespace(max_depth=20.0, max_depth_cov=0.2, mask_width=32))
    torch.manual_seed(7)
    px = s3.select_point(frame, 500, dep0, dep1, match)
    torch.manual_seed(7)
    assert torch.equal(px.cpu(), selector.mapping_point_selector(d0, d0c, 500, 20.0, 0.2, 32)[0])
