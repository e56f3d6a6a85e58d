"""CPU: the FlowFormerCov-shaped host network (tools/flowformer_host.py — measurement plumbing for the end-to-end leg, not part of the
drop-in).  The three methods the HIP library replaces are written there as the published definitions; here they are pinned to the oracle
(oracle/corr.py, oracle/frontend.py), so the UNHOOKED network is a valid "before" for the hooked one (tests/test_gpu_flowformer_host.py).
Weights are random and the absent submodule's arithmetic cannot be checked: parity unpinned, stated in the tool's header."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def host():
    import flowformer_host as fh

    torch.manual_seed(3)
    return fh, fh.FlowFormerCovHost(fh.demo_cfg(decoder_depth=2)).eval()


def test_published_methods_equal_the_oracle(host):
    from oracle import corr as ocorr
    from oracle import frontend as ofront

    fh, m = host
    g = torch.Generator().manual_seed(0)
    f1, f2 = torch.randn(2, 256, 6, 8, generator=g), torch.randn(2, 256, 6, 8, generator=g)
    vol = m.memory_encoder.corr(f1, f2)
    assert vol.shape == (2, 1, 6, 8, 6, 8)
    ref = ocorr.corr_volume(f1, f2)                                         # [B * H1 * W1, 1, H2, W2]
    torch.testing.assert_close(vol.permute(0, 2, 3, 1, 4, 5).reshape(96, 1, 6, 8), ref, rtol=1e-5, atol=1e-4)
    coords = ocorr.coords_grid(2, 6, 8) + (torch.rand(2, 2, 6, 8, generator=g) - 0.5) * 6
    tok = m.memory_decoder.encode_flow_token(ref, coords)
    assert tok.shape == (2, 81, 6, 8)
    torch.testing.assert_close(tok, ocorr.corr_lookup(ref, coords, 4), rtol=1e-6, atol=1e-6)
    flow, mask = torch.randn(2, 2, 6, 8, generator=g), torch.randn(2, 576, 6, 8, generator=g)
    torch.testing.assert_close(m.memory_decoder.upsample_flow(flow, mask), ofront.upsample_flow(flow, mask), rtol=1e-6, atol=1e-6)


def test_network_runs_and_has_the_layout_the_hooks_expect(host):
    """One tiny forward (shapes, finite values, determinism) and the attribute layout `plugins.install_flowformer_hooks` walks: the three
    methods + the cost patch embedding's `proj` stack of three 6x6 stride-2 convolutions (flownet.py:18-44, covhead.py:60-140)."""
    fh, m = host
    g = torch.Generator().manual_seed(1)
    a, b = torch.rand(2, 3, 64, 96, generator=g), torch.rand(2, 3, 64, 96, generator=g)
    with torch.no_grad():
        flow, cov = m.inference(a, b)
        flow2, cov2 = m.inference(a, b)
    assert flow.shape == (2, 2, 64, 96) and cov.shape == (2, 2, 64, 96)
    assert torch.isfinite(flow).all() and torch.isfinite(cov).all() and (cov > 0).all()
    assert torch.equal(flow, flow2) and torch.equal(cov, cov2)
    assert callable(m.memory_encoder.corr) and callable(m.memory_decoder.encode_flow_token) and callable(m.memory_decoder.upsample_flow)
    pe = dict(m.memory_encoder.named_modules())["cost_perceiver_encoder.patch_embed"]
    convs = [c for c in pe.proj if isinstance(c, torch.nn.Conv2d)]
    assert [(c.in_channels, c.out_channels, c.kernel_size, c.stride, c.padding) for c in convs] == [
        (1, 16, (6, 6), (2, 2), (2, 2)), (16, 32, (6, 6), (2, 2), (2, 2)), (32, 64, (6, 6), (2, 2), (2, 2))]
    assert 15e6 < fh.parameter_count(m) < 25e6          # FlowFormer (latentcostformer, Twins-SVT-L stem x 2): ~18 M parameters


def test_non_multiple_of_eight_images_are_padded_and_unpadded(host):
    fh, m = host
    a, b = torch.rand(1, 3, 60, 90), torch.rand(1, 3, 60, 90)
    with torch.no_grad():
        flow, cov = m.inference(a, b)
    assert flow.shape == (1, 2, 60, 90) and cov.shape == (1, 2, 60, 90)
