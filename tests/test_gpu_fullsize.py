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
