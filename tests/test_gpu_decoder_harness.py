"""The decoder-loop harness (tools/decoder_harness.py): HIP lookups / upsamplings issued between real PyTorch-ROCm
kernels on one stream, in the statement order of MemoryCovDecoder.forward (covhead.py:85-135)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dec_dtype", [torch.float32, torch.bfloat16])
def test_interleaved_lookups_and_upsampling_match_the_oracle(gpu, dec_dtype):
    from macvo_amd import ops
    from tools.decoder_harness import DecoderLoopHarness
    from oracle import corr, frontend

    torch.manual_seed(0)
    B, C, h8, w8, depth = 2, 64, 24, 32, 4
    f1, f2 = torch.randn(B, C, h8, w8), torch.randn(B, C, h8, w8)
    vol = ops.corr_volume(f1.to(gpu), f2.to(gpu))
    net = DecoderLoopHarness(dec_dtype=dec_dtype, depth=depth).to(gpu).eval()
    net.trace = []
    memory = torch.randn(B * h8 * w8, 8, 128, device=gpu)
    context = torch.randn(B, 256, h8, w8, device=gpu)
    (flow_up, flow8), (cov_up, cov8) = net(vol, memory, context, time_hip=True)
    torch.cuda.synchronize()
    assert flow_up.shape == (B, 2, 8 * h8, 8 * w8) and cov_up.shape == flow_up.shape
    assert torch.isfinite(flow_up).all() and torch.isfinite(cov_up).all() and (cov_up > 0).all()      # exp(2 x) fused on the last iteration
    assert len(net.trace) == depth
    vol_cpu = vol.cpu()
    for it, tr in enumerate(net.trace):
        # every lookup saw the coordinates the previous iteration's GRU produced (not stale, not early)
        want = corr.corr_lookup(vol_cpu, tr["coords"].cpu(), 4)
        torch.testing.assert_close(tr["tokens"].cpu(), want, rtol=1e-5, atol=2e-4)
        up = frontend.upsample_flow(tr["flow8"].cpu(), 0.25 * tr["up_mask"].float().cpu())       # the kernel read the bf16 mask as it is
        torch.testing.assert_close(tr["flow_up"].cpu(), up, rtol=2e-5, atol=2e-5)
        if it:
            assert not torch.equal(tr["coords"], net.trace[it - 1]["coords"])
    t = net.hip_times_us()
    assert set(t) == {"corr_lookup", "convex_upsample"} and all(0 < v < 5e4 for v in t.values())
    assert torch.equal(net.last["flow8"], flow8) and net.last["up_mask"].shape == (B, 576, h8, w8)


def test_harness_outputs_drive_the_hot_path(gpu):
    """volume -> 12-iteration decoder loop (stand-in network) -> the native frame driver, frame after frame: the path
    `estimate_pair` -> selector -> covariance -> PGO runs end to end on network-shaped inputs (finite poses, 200 keypoints)."""
    from macvo_amd import ops
    from tools.decoder_harness import DecoderLoopHarness
    from macvo_amd.pipeline import Camera, FrameInputs, HotPathConfig, NativeHotPath
    from tests import synth

    torch.manual_seed(1)
    H, W, C, depth = 192, 256, 64, 3
    h8, w8 = H // 8, W // 8
    cam = synth.make_camera(H, W)
    net = DecoderLoopHarness(dec_dtype=torch.bfloat16, depth=depth).to(gpu).eval()
    hot = NativeHotPath(Camera(**cam), HotPathConfig(), gpu)
    poses = []
    for t in range(4):
        f1, f2 = torch.randn(2, C, h8, w8, device=gpu), torch.randn(2, C, h8, w8, device=gpu)
        vol = ops.corr_volume(f1, f2)
        net.trace = []
        net(vol, torch.randn(2 * h8 * w8, 8, 128, device=gpu), torch.randn(2, 256, h8, w8, device=gpu))
        last = net.last
        # the stand-in network is random: give the stereo pair a usable disparity and keep log-sigma moderate
        last["flow8"][0, 0] = -(2.0 + last["flow8"][0, 0].abs().clamp(max=2.0))
        last["cov8"].clamp_(-1.0, 1.0)
        coords = torch.stack([tr["coords"] for tr in net.trace])
        x = FrameInputs(fmap1=f1, fmap2=f2, coords=coords, **last)
        torch.cuda.synchronize()
        if t == 0:
            hot.initialize(x)
            continue
        r = hot.step(x)
        torch.cuda.synchronize()
        torch.testing.assert_close(hot.last_tokens, net.trace[-1]["tokens"], rtol=0, atol=0)   # same lookup, inside and outside the loop
        assert r.n_cand > 0 and torch.isfinite(r.pose).all()
        poses.append(r.pose.clone())
    assert len(poses) == 3
