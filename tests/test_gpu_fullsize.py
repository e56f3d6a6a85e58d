"""GPU: BASELINE.json full-size configurations — configs[2] (1280x720 -> 90x160, N = 14400) and configs[4]
(batch-32 frames/GPU -> B = 64 pairs at 60x80).  The CPU oracle is too slow / too big for whole-tensor comparison at these
sizes, so parity is checked (i) exactly against the oracle on sampled rows / slices and (ii) through size-independent
properties: bilinearity of the volume, integer-shift equivariance of the lookup, batch independence."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _feats(B, C, H, W, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, C, H, W, generator=g).to(dtype), torch.randn(B, C, H, W, generator=g).to(dtype)


@pytest.mark.parametrize("B,H,W", [(2, 90, 160), (64, 60, 80)])
def test_volume_fullsize_sampled_rows_and_linearity(gpu, B, H, W):
    from macvo_amd import ops

    C = 256
    f1, f2 = _feats(B, C, H, W, seed=0)
    N = H * W
    d1, d2 = f1.to(gpu), f2.to(gpu)
    vol = ops.corr_volume(d1, d2)
    assert vol.shape == (B * N, 1, H, W)
    # (i) sampled query rows vs float64 dot products
    g = torch.Generator().manual_seed(1)
    for _ in range(6):
        b = int(torch.randint(0, B, (1,), generator=g))
        q = int(torch.randint(0, N, (1,), generator=g))
        ref = (f1[b].reshape(C, N)[:, q].double()[:, None] * f2[b].reshape(C, N).double()).sum(0)
        got = vol[b * N + q].reshape(-1).cpu().double()
        assert (got - ref).abs().max().item() <= 2e-5 * C ** 0.5
    # edge tiles: last query row / last column block
    for b, q in ((0, N - 1), (B - 1, N - 1), (B - 1, 0)):
        ref = (f1[b].reshape(C, N)[:, q].double()[:, None] * f2[b].reshape(C, N).double()).sum(0)
        assert (vol[b * N + q].reshape(-1).cpu().double() - ref).abs().max().item() <= 2e-5 * C ** 0.5
    # (ii) homogeneity in f1 with a power-of-two scale is exact in fp32
    chk = vol[: 4 * N : 997].clone()
    vol2 = ops.corr_volume(d1 * 4.0, d2)
    assert torch.equal(vol2[: 4 * N : 997], chk * 4.0)
    del vol2
    # (iii) batch independence: recomputing a single pair alone reproduces its slices bit for bit
    b = B - 1
    solo = ops.corr_volume(d1[b:b + 1].contiguous(), d2[b:b + 1].contiguous())
    assert torch.equal(solo, vol[b * N:(b + 1) * N])


@pytest.mark.parametrize("B,H,W", [(2, 90, 160), (64, 60, 80)])
def test_lookup_fullsize_slices_and_shift(gpu, B, H, W):
    from macvo_amd import ops
    from oracle import corr

    N = H * W
    g = torch.Generator().manual_seed(3)
    # a synthetic volume with per-slice structure (cheap to build on the GPU, float32)
    vol = torch.randn(B * N, 1, H, W, generator=g, dtype=torch.float32) if B * N * N < 3e8 else None
    if vol is None:
        vol_d = torch.randn(B * N, 1, H, W, device=gpu, dtype=torch.float32)
    else:
        vol_d = vol.to(gpu)
    coords = corr.coords_grid(B, H, W) + (torch.rand(B, 2, H, W, generator=g) * 2 - 1) * 8
    out = ops.corr_lookup(vol_d, coords.to(gpu), 4)
    assert out.shape == (B, 81, H, W)
    # (i) oracle on sampled batch items (grid_sample over one item's N slices)
    for b in (0, B - 1):
        sl = vol_d[b * N:(b + 1) * N].cpu()
        ref = corr.corr_lookup(sl, coords[b:b + 1], 4)
        torch.testing.assert_close(out[b:b + 1].cpu(), ref, rtol=1e-5, atol=2e-4)
    # (ii) shifting every coordinate by +1 in x moves window column i to i+1 (channels 9*i + j)
    out_s = ops.corr_lookup(vol_d, (coords + torch.tensor([1.0, 0.0]).view(1, 2, 1, 1)).to(gpu), 4)
    a = out.view(B, 9, 9, H, W)[:, 1:]      # taps x + (i - 4), i = 1..8
    b_ = out_s.view(B, 9, 9, H, W)[:, :-1]  # taps (x + 1) + (i - 4), i = 0..7
    torch.testing.assert_close(b_, a, rtol=1e-4, atol=2e-3)


_STREAM16_720P = r"""
import hashlib, sys, torch
sys.path.insert(0, sys.argv[1])
from macvo_amd import ops
for dt in (torch.float16, torch.bfloat16):
    g = torch.Generator().manual_seed(21)
    f1 = torch.randn(2, 90, 160, 256, generator=g).to(dt).cuda()
    f2 = torch.randn(2, 90, 160, 256, generator=g).to(dt).cuda()
    out = ops.corr_volume(f1, f2, layout="hwc")
    torch.cuda.synchronize()
    h = hashlib.sha1()
    for b in range(2):                                   # 1.66 GB: hash it pair by pair
        h.update(out[b * 14400:(b + 1) * 14400].cpu().numpy().tobytes())
    print(ops.last_volume_kernel(), h.hexdigest())
"""


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_volume_16bit_streaming_kernel_at_720p(gpu, dtype):
    """BASELINE configs[2] as the Fast-mode bench line runs it (Config/Experiment/MACVO/MACVO_Fast.yaml:69-76: 16-bit encoder
    features): 1280x720 -> N = 14400 queries, C = 256, HWC, B = 2 — the shape at which `corr_volume_h_stream` changes its XCD
    region count (4 regions instead of 2).  (i) the dispatcher must pick the streaming kernel; (ii) sampled query rows, the
    edge rows of the half band at the bottom (14400 = 112.5 x 128) and the last column block vs fp64 dot products of the same
    rounded features; (iii) exact homogeneity under a power-of-two scale."""
    from macvo_amd import ops

    B, H, W, C = 2, 90, 160, 256
    N = H * W
    g = torch.Generator().manual_seed(21)
    f1 = torch.randn(B, H, W, C, generator=g).to(dtype)
    f2 = torch.randn(B, H, W, C, generator=g).to(dtype)
    d1, d2 = f1.to(gpu), f2.to(gpu)
    vol = ops.corr_volume(d1, d2, layout="hwc")
    assert ops.last_volume_kernel() == "corr_volume_h_stream"
    assert vol.shape == (B * N, 1, H, W)
    a, b_ = f1.reshape(B, N, C).double(), f2.reshape(B, N, C).double()
    gq = torch.Generator().manual_seed(2)
    rows = [(int(torch.randint(0, B, (1,), generator=gq)), int(torch.randint(0, N, (1,), generator=gq))) for _ in range(8)]
    rows += [(0, 0), (0, N - 1), (B - 1, N - 1), (B - 1, 14336), (1, 14335), (0, 127), (0, 128), (1, N - 64), (1, N - 65)]
    for b, q in rows:
        ref = b_[b] @ a[b, q]
        got = vol[b * N + q].reshape(-1).cpu().double()
        assert (got - ref).abs().max().item() <= 2e-5 * C ** 0.5, (b, q)
    # every 601st row of the whole volume in one go (covers all bands / regions / both pairs)
    idx = torch.arange(0, B * N, 601)
    ref = torch.stack([b_[int(i) // N] @ a[int(i) // N, int(i) % N] for i in idx])
    assert (vol[idx.to(gpu)].reshape(len(idx), -1).cpu().double() - ref).abs().max().item() <= 2e-5 * C ** 0.5
    chk = vol[:: 997].clone()
    vol2 = ops.corr_volume(d1 * 4.0, d2, layout="hwc")
    assert torch.equal(vol2[:: 997], chk * 4.0)


def test_volume_16bit_streaming_equals_tile_form_at_720p(gpu):
    """... and its bits are those of the 128x128 tile kernel at this size too (MV_H_STREAM is read once per process, so each
    form runs in its own interpreter; the dispatch is asserted through mv_corr_volume_last_kernel)."""
    import os, subprocess, sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for flag in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", _STREAM16_720P, root], env=dict(os.environ, MV_H_STREAM=flag),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln.split() for ln in r.stdout.strip().splitlines()])
    assert [o[0] for o in outs[0]] == ["corr_volume_h_stream"] * 2 and [o[0] for o in outs[1]] == ["corr_volume_h_hwc"] * 2, outs
    assert [o[1] for o in outs[0]] == [o[1] for o in outs[1]], outs


def test_volume_streaming_gate_on_32bit_output_offsets(gpu):
    """The streaming kernels address a pair's output block with 32-bit byte offsets; the dispatcher must fall back to the tile
    kernel once N1 * N2 reaches 2^30 elements (corr_volume.hip, `mv_corr_volume`).  N = 32768 (a 1024 x 2048 image at 1/8
    resolution), C = 128, one pair: 4.3 GB of output.  Just below the gate (N1 = 32704) the streaming kernel must still run."""
    from macvo_amd import ops

    C = 128
    g = torch.Generator().manual_seed(5)
    for N, want in ((32768, "corr_volume_h_hwc"), (32704, "corr_volume_h_stream")):
        f1 = torch.randn(1, N, C, generator=g).to(torch.float16)
        f2 = torch.randn(1, 32768, C, generator=g).to(torch.float16)
        vol = ops.corr_volume(f1.to(gpu), f2.to(gpu), layout="hwc")
        assert ops.last_volume_kernel() == want, (N, ops.last_volume_kernel())
        a, b_ = f1[0].double(), f2[0].double()
        for q in (0, 1, 12345, N - 129, N - 1):
            got = vol[q].reshape(-1).cpu().double()
            assert (got - b_ @ a[q]).abs().max().item() <= 2e-5 * C ** 0.5, (N, q)
        # the far end of the block is where a wrapped 32-bit offset would have landed somewhere else
        last = vol[N - 1].reshape(-1)[-64:].cpu().double()
        assert (last - (b_[-64:] @ a[N - 1])).abs().max().item() <= 2e-5 * C ** 0.5
        del vol
        torch.cuda.empty_cache()
