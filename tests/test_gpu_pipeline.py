"""GPU parity of the whole per-frame hot path (mac-vo_amd/pipeline.py) vs the CPU oracle pipeline, frame by frame."""
import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu


def _to_inputs(fr, dev):
    from macvo_amd.pipeline import FrameInputs

    x = FrameInputs(**{k: v.to(dev) for k, v in fr.items()})
    x.ready = torch.cuda.Event()
    x.ready.record()
    return x


@pytest.mark.parametrize("H,W,graph,selector", [(480, 640, "disp", "nodepth"), (240, 320, "icp", "full"), (240, 320, "reproj", "nodepth")])
def test_sequence_matches_oracle(gpu, H, W, graph, selector):
    from macvo_amd.pipeline import Camera, HotPath, HotPathConfig
    from oracle import se3
    from oracle.pipeline import OracleHotPath

    n_frames = 5
    cam, frames, true_poses = synth.make_sequence(n_frames, H, W, C=64, iters=3, seed=3)
    cfg = dict(graph_type=graph, selector=selector)
    ora = OracleHotPath(cam, cfg)
    hot = HotPath(Camera(**cam), HotPathConfig(graph_type=graph, selector=selector), gpu, keep_extras=True)
    ora.initialize(frames[0])
    hot.initialize(_to_inputs(frames[0], gpu))
    for t in range(1, n_frames):
        torch.manual_seed(100 + t)
        ro = ora.step(frames[t])
        torch.manual_seed(100 + t)
        rh = hot.step(_to_inputs(frames[t], gpu))
        # lookup tokens of the last decoder iteration
        torch.testing.assert_close(hot.last_tokens.cpu(), ora.last_tokens, rtol=1e-5, atol=2e-4)
        # bit-exact keypoints
        assert torch.equal(rh.kp0_uv.cpu(), ro["kp0_uv"]), f"frame {t}: keypoints differ"
        assert int(rh.n_valid.item()) == ro["n_valid"]
        ex = rh.extras
        inb = ex["tracked"].inbound.cpu()
        torch.testing.assert_close(ex["cov0"].cpu()[inb], ro["cov0"], rtol=2e-4, atol=1e-7)
        torch.testing.assert_close(ex["cov1"].cpu()[inb], ro["cov1"], rtol=2e-4, atol=1e-7)
        torch.testing.assert_close(ex["pos_Tw"].cpu()[inb], ro["pos_Tw"], rtol=1e-6, atol=1e-6)
        # pose within the north_star tolerance after the same iteration count
        dt, dr = se3.pose_error(ro["pose"].double(), rh.pose.cpu().double())
        assert dt <= 1e-4 and dr <= 1e-4, (t, dt, dr)
        assert int(rh.info[0, 1].item()) == ro["steps"], (t, rh.info.cpu(), ro["steps"])
        # the chained poses keep tracking the truth (sanity of the synthetic stream, not a parity claim)
        et, er = se3.pose_error(true_poses[t], rh.pose.cpu().double())
        assert et < 0.05 and er < 0.01, (t, et, er)
        # keep both pipelines on the same prior so that fp32-level covariance differences cannot compound
        hot.pose = ro["pose"].to(gpu)


def test_pipelined_run_equals_stepwise(gpu):
    """HotPath.run (cross-frame software pipelining + side-stream PGO) must give exactly the poses of step()."""
    from macvo_amd.pipeline import Camera, HotPath, HotPathConfig

    n_frames = 7
    cam, frames, _ = synth.make_sequence(n_frames, 240, 320, C=64, iters=2, seed=5)
    ins = [_to_inputs(f, gpu) for f in frames]
    a = HotPath(Camera(**cam), HotPathConfig(), gpu)
    a.initialize(ins[0])
    torch.manual_seed(77)
    poses_a = []
    for t in range(1, n_frames):
        poses_a.append(a.step(ins[t]).pose.clone())
    b = HotPath(Camera(**cam), HotPathConfig(), gpu)
    b.initialize(ins[0])
    sink = torch.zeros(n_frames - 1, 7, device=gpu)
    torch.manual_seed(77)
    kps = [r.kp0_uv for r in b.run(ins[1:], pose_sink=sink)]
    torch.cuda.synchronize()
    assert len(kps) == n_frames - 1
    assert torch.equal(sink, torch.stack(poses_a))


def test_sequence_with_convex_upsample_path(gpu):
    """§8(f) rank 1 widened path: the hot path receives 1/8-res flow / log-sigma + masks and upsamples them itself."""
    from macvo_amd.pipeline import Camera, HotPath, HotPathConfig
    from oracle import se3
    from oracle.pipeline import OracleHotPath

    H, W, n_frames = 240, 320, 4
    cam, frames, _ = synth.make_sequence(n_frames, H, W, C=64, iters=2, seed=9)
    g = torch.Generator().manual_seed(2)
    for fr in frames:   # derive coarse fields whose upsampling is a smooth version of the dense ones
        fr["flow8"] = torch.nn.functional.avg_pool2d(fr["flow"], 8) / 8.0
        fr["cov8"] = torch.nn.functional.avg_pool2d(fr["logcov"], 8) / 8.0
        fr["up_mask"] = torch.randn(2, 576, H // 8, W // 8, generator=g)
        fr["cov_mask"] = torch.randn(2, 576, H // 8, W // 8, generator=g) * 0.25
        fr["flow"] = None
        fr["logcov"] = None
    ora = OracleHotPath(cam, {})
    hot = HotPath(Camera(**cam), HotPathConfig(), gpu)
    def dv(fr):
        from macvo_amd.pipeline import FrameInputs

        x = FrameInputs(**{k: (None if v is None else v.to(gpu)) for k, v in fr.items()})
        torch.cuda.synchronize()
        return x
    from macvo_amd import ops as _ops

    def exact_fields(fr):
        """The oracle consumes the HIP path's own upsampled flow / exp(2*cov) maps (bit-identical inputs to everything behind
        the upsampling): expf differs by an ulp between implementations, which can flip an exact-equality NMS tie — the
        upsampling itself is checked against the oracle in test_gpu_corr::test_convex_upsample (3e-6)."""
        flow = _ops.convex_upsample(fr["flow8"].to(gpu), fr["up_mask"].to(gpu), mask_scale=0.25)
        cov = _ops.convex_upsample(fr["cov8"].to(gpu), fr["cov_mask"].to(gpu), mask_scale=1.0, exp2_out=True)
        ref = OracleHotPath(cam, {}).frontend(fr)          # oracle upsampling, for the closeness check only
        torch.testing.assert_close(cov.cpu()[1:2], ref["flow_cov"][:, :2], rtol=1e-5, atol=0)
        out = dict(fr)
        out.update(flow=flow.cpu(), cov_exp=cov.cpu(), flow8=None)
        return out

    ora.initialize(exact_fields(frames[0]))
    hot.initialize(dv(frames[0]))
    for t in range(1, n_frames):
        torch.manual_seed(40 + t)
        ro = ora.step(exact_fields(frames[t]))
        torch.manual_seed(40 + t)
        rh = hot.step(dv(frames[t]))
        assert torch.equal(rh.kp0_uv.cpu(), ro["kp0_uv"]), f"frame {t}: keypoints differ"
        dt, dr = se3.pose_error(ro["pose"].double(), rh.pose.cpu().double())
        assert dt <= 1e-4 and dr <= 1e-4, (t, dt, dr)       # free-running: each pipeline chains on its own pose


def test_graph_replay_equals_eager(gpu):
    """use_graphs: the decoder-side segment replayed as a hipGraph must give the same keypoints and poses as eager."""
    from macvo_amd.pipeline import Camera, FrameInputs, HotPath, HotPathConfig

    n_pool, n_steps = 6, 14
    cam, frames, _ = synth.make_sequence(n_pool, 240, 320, C=64, iters=3, seed=11, closed_loop=True)
    ins = [FrameInputs(static=True, **{k: v.to(gpu) for k, v in f.items()}) for f in frames]
    torch.cuda.synchronize()
    outs = []
    for use_graphs in (False, True):
        hot = HotPath(Camera(**cam), HotPathConfig(use_graphs=use_graphs), gpu)
        hot.initialize(ins[0])
        sink = torch.zeros(n_steps, 7, device=gpu)
        torch.manual_seed(5)
        kps = [r.kp0_uv.clone() for r in hot.run((ins[(1 + k) % n_pool] for k in range(n_steps)), pose_sink=sink)]
        torch.cuda.synchronize()
        outs.append((sink.clone(), kps))
        if use_graphs:
            assert len(hot._graphs) == n_pool
    assert torch.equal(outs[0][0], outs[1][0])
    assert all(torch.equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))


def test_dense_mapping_tail_matches_oracle(gpu):
    """`mapping: true` (MACVO_Fast.yaml): MappingPointSelector(2000) on the previous frame's depth + the covariance model on
    those points + world positions (MACVO.py:313-337).  Selected map pixels must be the oracle's bit for bit (second
    randperm of the frame on the shared CPU generator), values to fp32 rounding; keypoints / poses unchanged."""
    from macvo_amd.pipeline import Camera, FrameInputs, HotPath, HotPathConfig
    from oracle.pipeline import OracleHotPath

    H, W, n_frames = 240, 320, 4
    cam, frames, _ = synth.make_sequence(n_frames, H, W, C=64, iters=1, seed=23)
    g = torch.Generator().manual_seed(1)
    for fr in frames:
        fr["image"] = torch.rand(1, 3, H, W, generator=g)
    # the synthetic plane sits at ~12 m with sigma_z^2 ~ 1e-2: widen the selector's gates so that it has candidates
    mcfg = dict(mapping=True, map_max_depth=13.0, map_max_depth_cov=0.5, map_num_point=500)
    ora = OracleHotPath(cam, mcfg)
    hot = HotPath(Camera(**cam), HotPathConfig(**mcfg), gpu)
    ins = [FrameInputs(**{k: v.to(gpu) for k, v in fr.items()}) for fr in frames]
    torch.cuda.synchronize()
    ora.initialize(frames[0])
    hot.initialize(ins[0])
    for t in range(1, n_frames):
        torch.manual_seed(70 + t)
        ro = ora.step(frames[t])
        torch.manual_seed(70 + t)
        rh = hot.step(ins[t])
        torch.cuda.synchronize()
        assert torch.equal(rh.kp0_uv.cpu(), ro["kp0_uv"])
        m, mo = rh.map_points, ro["map"]
        assert m is not None and m.uv.shape[0] == 500
        assert torch.equal(m.uv.cpu().long(), mo["uv"]), t                       # identical map pixels
        assert torch.equal(m.depth.cpu(), mo["depth"])
        torch.testing.assert_close(m.sigma_dd.cpu(), mo["sigma_dd"], rtol=1e-5, atol=0)   # exp(2*cov): expf ulp on the device
        torch.testing.assert_close(m.pos_Tc.cpu(), mo["pos_Tc"], rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(m.pos_Tw.cpu(), mo["pos_Tw"], rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(m.cov_Tc.cpu(), mo["cov_Tc"], rtol=2e-4, atol=1e-7)
        assert torch.equal(m.color.cpu(), mo["color"])
        hot.pose = ro["pose"].to(gpu)
