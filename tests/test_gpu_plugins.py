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
    dt, dr = se3.pose_error(ref.pose, got.double())          # float32 write-back of a float64 solve that matches to ~1e-8
    assert dt <= 2e-7 and dr <= 2e-7, (dt, dr)
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
    torch.testing.assert_close(out, ref, rtol=2e-4, atol=1e-7)
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


def test_frontend_plugin_matches_reference_records(gpu):
    """HIP_FlowFormerCovFrontend with an injected network stub: estimate_pair / estimate_depth / estimate_triplet must
    return what FlowFormerCovFrontend.inference_2_depth / _2_match (Frontend.py:183-200) would, bit for bit."""
    from types import SimpleNamespace

    from macvo_amd import plugins
    from oracle import frontend as ofr

    H, W = 64, 96
    g = torch.Generator().manual_seed(0)

    class Net:
        def __init__(self):
            self.calls = []

        def inference(self, a, b):
            self.calls.append((a.clone(), b.clone()))
            n = a.shape[0]
            gg = torch.Generator().manual_seed(100 + n)
            flow = (torch.randn(n, 2, H, W, generator=gg) * 8).to(a.device)
            flow[0, 0, 0, :5] = torch.tensor([0.0, -0.0, 1e-30, -3.5, 2.0])       # zero / negative disparities
            cov = torch.exp(2 * (0.5 * torch.randn(n, 2, H, W, generator=gg) - 0.7)).to(a.device)
            return flow.half(), cov.half()                                       # the plugin must .float() them

    def frame(seed):
        gg = torch.Generator().manual_seed(seed)
        return SimpleNamespace(imageL=torch.rand(1, 3, H, W, generator=gg), imageR=torch.rand(1, 3, H, W, generator=gg),
                               frame_baseline=0.25, fx=320.0)

    net = Net()
    cfg = SimpleNamespace(weight="", device="cuda", dec_dtype="fp32", enc_dtype="fp32", enforce_positive_disparity=True,
                          decoder_depth=12, model=net)
    fe = plugins.HIP_FlowFormerCovFrontend(cfg)
    assert fe.provide_cov == (True, True)
    f1, f2 = frame(1), frame(2)

    def check_depth(out, flow, cov, fr):
        d, dc, disp, dispc, bad = ofr.inference_2_depth(flow, cov, fr.frame_baseline, fr.fx, enforce_positive_disparity=True)
        assert torch.equal(out.disparity.cpu(), disp) and torch.equal(out.disparity_uncertainty.cpu(), dispc)
        assert torch.equal(out.depth.cpu(), d, ) and torch.equal(out.cov.cpu(), dc)
        assert torch.equal(out.mask.cpu(), bad)

    depth2, match = fe.estimate_pair(f1, f2)
    a, b = net.calls[-1]
    assert torch.equal(a.cpu(), torch.cat([f2.imageL, f1.imageL])) and torch.equal(b.cpu(), torch.cat([f2.imageR, f2.imageL]))
    flow, cov = [t.float().cpu() for t in Net().inference(a, b)]
    check_depth(depth2, flow[0:1], cov[0:1], f2)
    assert torch.equal(match.flow.cpu(), flow[1:2]) and torch.equal(match.cov.cpu()[:, :2], cov[1:2])
    assert match.cov.shape == (1, 3, H, W) and match.cov[:, 2].abs().max() == 0 and match.mask is None

    d1 = fe.estimate_depth(f1)
    flow, cov = [t.float().cpu() for t in Net().inference(*net.calls[-1])]
    check_depth(d1, flow[0:1], cov[0:1], f1)

    t1, t2, m12 = fe.estimate_triplet(f1, f2)
    a, b = net.calls[-1]
    assert a.shape[0] == 3 and torch.equal(b.cpu(), torch.cat([f1.imageL, f2.imageR, f2.imageL]))
    flow, cov = [t.float().cpu() for t in Net().inference(a, b)]
    check_depth(t1, flow[0:1], cov[0:1], f1)
    check_depth(t2, flow[1:2], cov[1:2], f2)
    assert torch.equal(m12.flow.cpu(), flow[2:3]) and torch.equal(m12.cov.cpu()[:, :2], cov[2:3])


def test_graph_frontend_plugin_replays_hip_lookups(gpu):
    """HIP_CUDAGraph_FlowFormerCovFrontend: the network stub calls the HIP window lookup (a ctypes launch on the current
    stream) among ordinary torch ops; captured once and replayed, it must give the eager plugin's outputs for new inputs."""
    from types import SimpleNamespace

    from macvo_amd import ops, plugins

    H, W = 64, 96
    h8, w8 = H // 8, W // 8

    class Net:
        def __init__(self):
            g = torch.Generator().manual_seed(0)
            self.vol = torch.randn(2 * h8 * w8, 1, h8, w8, generator=g).to(gpu)
            self.base = torch.stack(torch.meshgrid(torch.arange(w8), torch.arange(h8), indexing="xy")).float()[None].to(gpu)

        def inference(self, a, b):
            shift = torch.nn.functional.avg_pool2d(a - b, 8)[:, :2]                         # [2,2,h8,w8], depends on the inputs
            tok = ops.corr_lookup(self.vol, (self.base + shift).contiguous(), 4)           # HIP kernel inside the model
            flow = torch.nn.functional.interpolate(tok[:, :2], size=(H, W), mode="bilinear") * 4 + 1.5
            cov = torch.exp(torch.nn.functional.interpolate(tok[:, 2:4], size=(H, W), mode="bilinear").clamp(-3, 1))
            return flow, cov

    def frame(seed):
        gg = torch.Generator().manual_seed(seed)
        return SimpleNamespace(imageL=torch.rand(1, 3, H, W, generator=gg), imageR=torch.rand(1, 3, H, W, generator=gg),
                               frame_baseline=0.25, fx=320.0)

    mk = lambda cls: cls(SimpleNamespace(weight="", device="cuda", dec_dtype="fp32", enc_dtype="fp32",        # noqa: E731
                                         enforce_positive_disparity=False, decoder_depth=12, model=Net()))
    eager, graphed = mk(plugins.HIP_FlowFormerCovFrontend), mk(plugins.HIP_CUDAGraph_FlowFormerCovFrontend)
    frames = [frame(s) for s in range(5)]
    for t in range(1, 5):                                  # call 1 builds the graph, calls 2-4 replay it
        d_e, m_e = eager.estimate_pair(frames[t - 1], frames[t])
        d_g, m_g = graphed.estimate_pair(frames[t - 1], frames[t])
        torch.cuda.synchronize()
        assert torch.equal(d_e.depth, d_g.depth) and torch.equal(d_e.cov, d_g.cov), t
        assert torch.equal(m_e.flow, m_g.flow) and torch.equal(m_e.cov, m_g.cov), t
    assert graphed.cuda_graph is not None and tuple(graphed.cuda_graph.shape) == (2, 3, H, W)


def test_depth_and_matcher_plugins(gpu):
    """HIP_FlowFormerCovDepth / HIP_FlowFormerCovMatcher (the FrontendCompose pieces) vs the reference formulas."""
    from types import SimpleNamespace

    from macvo_amd import plugins
    from macvo_amd.interfaces import IMatcher, IStereoDepth
    from oracle import frontend as ofr

    H, W = 48, 64

    class Net:
        def inference(self, a, b):
            gg = torch.Generator().manual_seed(int(a.sum().item() * 1000) % 1000)
            flow = (torch.randn(a.shape[0], 2, H, W, generator=gg) * 6).to(a.device)
            cov = torch.exp(torch.randn(a.shape[0], 2, H, W, generator=gg)).to(a.device)
            return flow, cov

    def frame(seed):
        gg = torch.Generator().manual_seed(seed)
        return SimpleNamespace(imageL=torch.rand(1, 3, H, W, generator=gg), imageR=torch.rand(1, 3, H, W, generator=gg),
                               frame_baseline=0.25, fx=320.0)

    cfg = lambda: SimpleNamespace(weight="", device="cuda", enc_dtype="fp32", dec_dtype="fp32", model=Net())  # noqa: E731
    assert IStereoDepth.get_class("HIP_FlowFormerCovDepth") is plugins.HIP_FlowFormerCovDepth
    assert IMatcher.get_class("HIP_FlowFormerCovMatcher") is plugins.HIP_FlowFormerCovMatcher
    dep, mat = plugins.HIP_FlowFormerCovDepth(cfg()), plugins.HIP_FlowFormerCovMatcher(cfg())
    f1, f2 = frame(1), frame(2)
    out = dep.estimate(f1)
    flow, cov = [t.cpu() for t in Net().inference(f1.imageL.to(gpu), f1.imageR.to(gpu))]
    d, dc, disp, dispc, _ = ofr.inference_2_depth(flow, cov, 0.25, 320.0)
    assert torch.equal(out.depth.cpu(), d) and torch.equal(out.cov.cpu(), dc)
    assert torch.equal(out.disparity.cpu(), disp) and torch.equal(out.disparity_uncertainty.cpu(), dispc) and out.mask is None
    m = mat.estimate(f1, f2)
    flow, cov = [t.cpu() for t in Net().inference(f1.imageL.to(gpu), f2.imageL.to(gpu))]
    assert torch.equal(m.flow.cpu(), flow) and torch.equal(m.cov.cpu(), ofr.from_partial_cov(cov)) and m.mask is None
    assert dep.provide_cov and mat.provide_cov


@pytest.mark.parametrize("enc_dtype", [torch.float32, torch.float16])
def test_flowformer_hooks_rebind_volume_lookup_and_upsampling(gpu, enc_dtype):
    """install_flowformer_hooks on a model SHAPED like FlowFormerCov (flownet.py:18-44): `memory_encoder.corr` builds the
    volume, the decoder loop calls `encode_flow_token` and twice `upsample_flow` per iteration (covhead.py:92,124-126,133-135).
    The stand-in's own methods are plain-torch restatements (einsum / grid_sample / unfold-softmax) that RAISE if they are still
    reached after the hooks went in, so every one of the three kernels provably runs through the HIP library; results are
    compared with the oracle's definitions."""
    from types import SimpleNamespace

    from macvo_amd import plugins
    from oracle import corr, frontend

    B, C, h8, w8, depth = 2, 64, 12, 16, 3
    g = torch.Generator().manual_seed(3)

    class Encoder:
        cfg = SimpleNamespace(cost_heads_num=1)

        def corr(self, fmap1, fmap2):
            raise AssertionError("MemoryEncoder.corr was not rebound")

        def __call__(self, fmap1, fmap2):
            c = self.corr(fmap1, fmap2)                                    # [B, heads, H1, W1, H2, W2], feature dtype
            assert c.shape == (B, 1, h8, w8, h8, w8) and c.dtype == fmap1.dtype
            return c.permute(0, 2, 3, 1, 4, 5).reshape(B * h8 * w8, 1, h8, w8)   # cost_maps as MemoryEncoder hands them on

    class Decoder:
        def encode_flow_token(self, cost_maps, coords):
            raise AssertionError("encode_flow_token was not rebound")

        def upsample_flow(self, flow, mask):
            raise AssertionError("upsample_flow was not rebound")

    class Net:
        def __init__(self):
            self.memory_encoder, self.memory_decoder = Encoder(), Decoder()
            self.trace = []

        def forward(self, fmap1, fmap2, deltas, masks, cmasks):
            cost_maps = self.memory_encoder(fmap1.to(enc_dtype), fmap2.to(enc_dtype)).float()      # flownet.py:26-27
            coords0 = corr.coords_grid(B, h8, w8).to(fmap1.device)
            coords1, cov1 = coords0.clone(), coords0.clone()
            for it in range(depth):
                tok = self.memory_decoder.encode_flow_token(cost_maps, coords1)                     # covhead.py:92
                coords1 = coords1 + deltas[it]
                flow_up = self.memory_decoder.upsample_flow(coords1 - coords0, 0.25 * masks[it])    # :124-126
                cov1 = cov1 + 0.1 * deltas[it]
                cov_up = self.memory_decoder.upsample_flow(cov1 - coords0, cmasks[it])              # :133-135
                self.trace.append(dict(coords=coords1 - deltas[it], tok=tok, flow8=coords1 - coords0, flow_up=flow_up,
                                       cov8=cov1 - coords0, cov_up=cov_up))
            return cost_maps

    net = Net()
    done = plugins.install_flowformer_hooks(net)
    assert done == ["memory_decoder.encode_flow_token", "memory_decoder.upsample_flow", "memory_encoder.corr"]
    f1, f2 = torch.randn(B, C, h8, w8, generator=g), torch.randn(B, C, h8, w8, generator=g)
    deltas = [torch.rand(B, 2, h8, w8, generator=g) * 4 - 2 for _ in range(depth)]
    masks = [torch.randn(B, 576, h8, w8, generator=g) for _ in range(depth)]
    cmasks = [torch.randn(B, 576, h8, w8, generator=g) for _ in range(depth)]
    dev = lambda xs: [x.to(gpu) for x in xs]  # noqa: E731
    cost_maps = net.forward(f1.to(gpu), f2.to(gpu), dev(deltas), dev(masks), dev(cmasks))
    torch.cuda.synchronize()
    # volume: fp64 einsum of the features as the encoder saw them (16-bit features: rounded first, result rounded once more)
    a, b = f1.to(enc_dtype).double(), f2.to(enc_dtype).double()
    want = corr.corr_volume(a, b, torch.float64)
    if enc_dtype == torch.float32:
        assert (cost_maps.cpu().double() - want).abs().max() <= 2e-5 * C ** 0.5
    else:
        torch.testing.assert_close(cost_maps.cpu(), want.to(enc_dtype).float(), rtol=2e-3, atol=2e-3)   # one fp16 rounding of ~|8|
    vol_cpu = cost_maps.cpu()
    for it, tr in enumerate(net.trace):
        torch.testing.assert_close(tr["tok"].cpu(), corr.corr_lookup(vol_cpu, tr["coords"].cpu(), 4), rtol=1e-5, atol=2e-4)
        torch.testing.assert_close(tr["flow_up"].cpu(), frontend.upsample_flow(tr["flow8"].cpu(), 0.25 * masks[it]), rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(tr["cov_up"].cpu(), frontend.upsample_flow(tr["cov8"].cpu(), cmasks[it]), rtol=2e-5, atol=2e-5)
    # a model that carries none of the three attributes is an error, not a silent no-op
    with pytest.raises(Exception):
        plugins.install_flowformer_hooks(SimpleNamespace())
