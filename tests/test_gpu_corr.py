"""GPU parity: HIP cost volume + window lookup vs the CPU oracle (oracle/corr.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _feats(B, C, H, W, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, C, H, W, generator=g).to(dtype), torch.randn(B, C, H, W, generator=g).to(dtype)


@pytest.mark.parametrize("shape", [(1, 64, 8, 12), (2, 256, 16, 24), (1, 32, 7, 9), (1, 256, 60, 80)])
def test_corr_volume_f32_chw(gpu, shape):
    from macvo_amd import ops
    from oracle import corr

    B, C, H, W = shape
    f1, f2 = _feats(B, C, H, W, seed=0)
    ref64 = corr.corr_volume(f1, f2, torch.float64)
    out = ops.corr_volume(f1.to(gpu), f2.to(gpu), layout="chw", precision="exact").cpu()
    assert out.shape == (B * H * W, 1, H, W) and out.dtype == torch.float32
    # fp32 fma chain over C products of N(0,1): tolerance 1e-5 relative to the row scale sqrt(C)
    scale = float(C) ** 0.5
    assert (out.double() - ref64).abs().max().item() <= 2e-5 * scale
    # and not worse than a float32 einsum on the CPU
    ref32 = corr.corr_volume(f1, f2, torch.float32)
    assert (out.double() - ref64).abs().max() <= 4 * (ref32.double() - ref64).abs().max() + 1e-6


def test_corr_volume_asymmetric_identity(gpu):
    """A = I-like probe with asymmetric B catches transposed C writes (guide §3)."""
    from macvo_amd import ops

    C, H, W = 32, 4, 8  # N = 32 = C
    N = H * W
    f1 = torch.eye(C).reshape(1, C, H, W).contiguous()          # f1[c, i] = delta(c, i)
    f2 = torch.arange(C * N, dtype=torch.float32).reshape(1, C, H, W) / 7.0
    out = ops.corr_volume(f1.to(gpu), f2.to(gpu), precision="exact").cpu().reshape(N, N)
    assert torch.equal(out, f2.reshape(C, N))                    # out[i, j] = f2[i, j]


@pytest.mark.parametrize("shape", [(2, 64, 8, 12), (1, 256, 60, 80), (1, 48, 5, 7)])
def test_corr_volume_f32_hwc(gpu, shape):
    from macvo_amd import ops
    from oracle import corr

    B, C, H, W = shape
    f1, f2 = _feats(B, C, H, W, seed=3)
    ref64 = corr.corr_volume(f1, f2, torch.float64)
    out = ops.corr_volume(f1.permute(0, 2, 3, 1).contiguous().to(gpu), f2.permute(0, 2, 3, 1).contiguous().to(gpu),
                          layout="hwc").cpu()
    assert (out.double() - ref64).abs().max().item() <= 2e-5 * float(C) ** 0.5


@pytest.mark.parametrize("shape", [(2, 64, 8, 12), (1, 256, 60, 80), (1, 96, 5, 7), (1, 256, 16, 24)])
def test_corr_volume_f32_split3(gpu, shape):
    """bf16x3 split mode: fp32 operands rebuilt from six bf16 MFMA products.  Same tolerance as the exact fp32 path
    (2e-5 * sqrt(C) absolute vs a float64 einsum) — i.e. fp32-class accuracy, far tighter than the TF32/fp16 the
    reference runs this GEMM in (Frontend.py:275-278)."""
    from macvo_amd import ops
    from oracle import corr

    B, C, H, W = shape
    f1, f2 = _feats(B, C, H, W, seed=5)
    f1[0, :, 0, 0] *= 1e3       # wide dynamic range within a row
    f2[0, :, 0, 1] *= 1e-3
    ref64 = corr.corr_volume(f1, f2, torch.float64)
    a1, a2 = f1.permute(0, 2, 3, 1).contiguous().to(gpu), f2.permute(0, 2, 3, 1).contiguous().to(gpu)
    out = ops.corr_volume(a1, a2, layout="hwc", precision="split3").cpu()
    err = (out.double() - ref64).abs()
    scale = (f1.double().abs().reshape(B, C, -1).permute(0, 2, 1).unsqueeze(2) * f2.double().abs().reshape(B, C, -1).permute(0, 2, 1).unsqueeze(1)).sum(-1)
    assert (err / scale.reshape(err.shape).clamp_min(1e-30)).max().item() <= 1e-6     # relative to sum |a||b|
    exact = ops.corr_volume(a1, a2, layout="hwc", precision="exact").cpu()
    e_exact = (exact.double() - ref64).abs()
    assert err.max() <= 1.5 * e_exact.max() + 1e-6                                     # no worse than the exact fp32 path


@pytest.mark.parametrize("shape", [(2, 64, 8, 12), (1, 256, 60, 80), (1, 96, 5, 7)])
def test_corr_volume_f32_split2(gpu, shape):
    """Two leading bf16 pieces, three products: the relative error (vs sum |a||b|) must stay below 2^-15 — 16x finer
    than TF32 (2^-11), the precision the reference's fast frontend runs this GEMM in (Frontend.py:275-277)."""
    from macvo_amd import ops
    from oracle import corr

    B, C, H, W = shape
    f1, f2 = _feats(B, C, H, W, seed=6)
    f1[0, :, 0, 0] *= 1e3
    f2[0, :, 0, 1] *= 1e-3
    ref64 = corr.corr_volume(f1, f2, torch.float64)
    a1, a2 = f1.permute(0, 2, 3, 1).contiguous().to(gpu), f2.permute(0, 2, 3, 1).contiguous().to(gpu)
    out = ops.corr_volume(a1, a2, layout="hwc", precision="split2").cpu()
    err = (out.double() - ref64).abs()
    scale = (f1.double().abs().reshape(B, C, -1).permute(0, 2, 1).unsqueeze(2) * f2.double().abs().reshape(B, C, -1).permute(0, 2, 1).unsqueeze(1)).sum(-1)
    rel = (err / scale.reshape(err.shape).clamp_min(1e-30)).max().item()
    assert rel <= 2.0 ** -15, rel
    # and it is a genuinely different (coarser) mode than split3
    fine = ops.corr_volume(a1, a2, layout="hwc", precision="split3").cpu()
    assert (fine.double() - ref64).abs().max() <= err.max()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("layout", ["chw", "hwc"])
@pytest.mark.parametrize("shape", [(2, 64, 8, 12), (1, 256, 60, 80), (1, 128, 9, 11), (2, 256, 59, 64), (2, 128, 48, 64)])
def test_corr_volume_16bit(gpu, dtype, layout, shape):
    """Fast-mode path: 16-bit operands, fp32 accumulate.  Oracle = float64 einsum of the SAME rounded inputs;
    tolerance = fp32 accumulation error only (1e-5 * sqrt(C) scale)."""
    from macvo_amd import ops
    from oracle import corr

    B, C, H, W = shape
    f1, f2 = _feats(B, C, H, W, seed=4, dtype=dtype)
    ref64 = corr.corr_volume(f1, f2, torch.float64)
    if layout == "chw":
        out = ops.corr_volume(f1.to(gpu), f2.to(gpu), layout="chw", precision="exact")
    else:
        out = ops.corr_volume(f1.permute(0, 2, 3, 1).contiguous().to(gpu), f2.permute(0, 2, 3, 1).contiguous().to(gpu),
                              layout="hwc")
    assert (out.cpu().double() - ref64).abs().max().item() <= 2e-5 * float(C) ** 0.5


_STREAM_VS_TILE = r"""
import hashlib, sys, torch
sys.path.insert(0, sys.argv[1])
from macvo_amd import ops
for dt, (B, C, H, W) in ((torch.float16, (2, 256, 60, 80)), (torch.bfloat16, (2, 256, 59, 64)), (torch.float16, (3, 128, 48, 64))):
    g = torch.Generator().manual_seed(11)
    f1 = torch.randn(B, H, W, C, generator=g).to(dt).cuda()
    f2 = torch.randn(B, H, W, C, generator=g).to(dt).cuda()
    out = ops.corr_volume(f1, f2, layout="hwc")
    print(hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest())
"""


def test_corr_volume_16bit_streaming_form_is_bitwise_the_tile_form(gpu):
    """The streaming kernel (whole-K A fragments in registers, LDS-DMA ring, hand-counted waits) must give the very bits of the
    128x128 tile kernel it replaces: same MFMA, same k order.  Shapes: full bands, a half band at the bottom edge, three pairs
    with C = 128.  MV_H_STREAM is read once per process, so each form runs in its own interpreter."""
    import os, subprocess, sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shas = []
    for flag in ("1", "0"):
        env = dict(os.environ, MV_H_STREAM=flag)
        r = subprocess.run([sys.executable, "-c", _STREAM_VS_TILE, root], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        shas.append(r.stdout.split())
    assert len(shas[0]) == 3 and shas[0] == shas[1], shas


_F32_STREAM_VS_TILES = r"""
import hashlib, sys, torch
sys.path.insert(0, sys.argv[1])
from macvo_amd import ops
for B, C, H, W in ((2, 256, 60, 80), (1, 256, 64, 64), (3, 256, 59, 64)):
    g = torch.Generator().manual_seed(12)
    f1 = torch.randn(B, C, H, W, generator=g).cuda()
    f2 = torch.randn(B, C, H, W, generator=g).cuda()
    out = ops.corr_volume(f1, f2, layout="chw", precision="exact")
    print(hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest())
"""


def test_corr_volume_f32_staging_variants_are_bitwise_equal(gpu):
    """The fp32 mixed-tile GEMM with LDS-DMA staging (BK 32 x 2 stages = default, BK 16 x 3 stages) against the register-staged
    tile it replaced: k ascends per output element in all of them, so the volumes must agree bit for bit."""
    import os, subprocess, sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shas = []
    for flag in ("32", "16", "4", "0"):
        env = dict(os.environ, MV_VOL_DMA=flag)
        r = subprocess.run([sys.executable, "-c", _F32_STREAM_VS_TILES, root], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        shas.append(r.stdout.split())
    assert len(shas[0]) == 3 and all(x == shas[0] for x in shas[1:]), shas


def _coords(B, H, W, seed, spread=8.0):
    from oracle import corr

    g = torch.Generator().manual_seed(seed)
    return corr.coords_grid(B, H, W) + (torch.rand(B, 2, H, W, generator=g) * 2 - 1) * spread


def test_corr_lookup_both_variants_agree_for_every_radius(gpu):
    """The two launch shapes of the lookup (16-query / 32-query workgroups; B * N1 <= / > MV_LOOKUP_SMALL) deal a wave's (tap, query) pairs differently
    (2 / 4 queries per wave, (2r+1)^2 taps each): for every radius the forced 4-queries-per-wave variant returns the default variant's tokens bit for bit
    — coordinates inside, across the border and far outside; ragged N1 (not a multiple of 32)."""
    import os, subprocess, sys

    code = (
        "import sys, torch; sys.path.insert(0, sys.argv[1])\n"
        "from macvo_amd import ops\n"
        "from oracle import corr\n"
        "g = torch.Generator().manual_seed(3)\n"
        "for r in (1, 2, 3, 4):\n"
        "    for B, H, W in ((2, 13, 17), (1, 24, 32)):\n"
        "        vol = (torch.randn(B * H * W, 1, H, W, generator=g) * 8).cuda()\n"
        "        co = corr.coords_grid(B, H, W) + torch.rand(B, 2, H, W, generator=g) * 14 - 7\n"
        "        co[:, :, 0, 0] = 500.0\n"
        "        t = ops.corr_lookup(vol, co.cuda(), r)\n"
        "        torch.save(t.cpu(), sys.argv[2] + f'/t_{r}_{B}_{H}.pt')\n"
    )
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile

    outs = []
    for thr in ("1000000000", "0"):
        d = tempfile.mkdtemp()
        r = subprocess.run([sys.executable, "-c", code, root, d], env=dict(os.environ, MV_LOOKUP_SMALL=thr), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append({f: torch.load(os.path.join(d, f)) for f in sorted(os.listdir(d))})
    assert outs[0].keys() == outs[1].keys() and len(outs[0]) == 8
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize("shape,radius", [((2, 8, 12), 4), ((1, 16, 24), 4), ((1, 7, 9), 3), ((2, 5, 6), 1), ((1, 9, 5), 2)])
def test_corr_lookup_small(gpu, shape, radius):
    from macvo_amd import ops
    from oracle import corr

    B, H, W = shape
    g = torch.Generator().manual_seed(10)
    vol = torch.randn(B * H * W, 1, H, W, generator=g) * 16
    coords = _coords(B, H, W, seed=1)
    ref = corr.corr_lookup(vol, coords, radius)
    out = ops.corr_lookup(vol.to(gpu), coords.to(gpu), radius).cpu()
    assert out.shape == ref.shape
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=2e-4)


def test_corr_lookup_integer_coords_is_direct_indexing(gpu):
    """At integer coordinates the lookup must reproduce direct indexing (with zero padding) up to the fp32
    normalise/un-normalise round trip of grid_sample (<= 1e-5 relative)."""
    from macvo_amd import ops
    from oracle import corr

    B, H, W, r = 1, 12, 16, 4
    g = torch.Generator().manual_seed(11)
    vol = torch.randn(B * H * W, 1, H, W, generator=g)
    coords = corr.coords_grid(B, H, W)
    out = ops.corr_lookup(vol.to(gpu), coords.to(gpu), r).cpu()
    K = 2 * r + 1
    exp = torch.zeros(B, K * K, H, W)
    for y in range(H):
        for x in range(W):
            q = y * W + x
            for i in range(K):
                for j in range(K):
                    xx, yy = x + i - r, y + j - r
                    if 0 <= xx < W and 0 <= yy < H:
                        exp[0, K * i + j, y, x] = vol[q, 0, yy, xx]
    torch.testing.assert_close(out, exp, rtol=1e-4, atol=1e-4)


def test_corr_lookup_full_size_12_iters(gpu):
    """BASELINE size: 640x480 -> 60x80, B = 2, 12 successive coordinate sets (seeds 1..12, ~5% taps outside)."""
    from macvo_amd import ops
    from oracle import corr

    B, H, W = 2, 60, 80
    f1, f2 = _feats(B, 256, H, W, seed=0)
    vol_d = ops.corr_volume(f1.to(gpu), f2.to(gpu))
    vol = vol_d.cpu()
    for it in range(1, 13):
        coords = _coords(B, H, W, seed=it)
        ref = corr.corr_lookup(vol, coords, 4)
        out = ops.corr_lookup(vol_d, coords.to(gpu), 4).cpu()
        torch.testing.assert_close(out, ref, rtol=1e-5, atol=2e-4)


@pytest.mark.parametrize("B", [1, 20])
def test_corr_lookup_margin_rows_at_integer_boundaries(gpu, B):
    """The lookup fetches the two margin rows / columns of its 12 x 12 block only when a coordinate's fraction is near 0 or 1 (where the
    fp32 normalise / un-normalise round trip of grid_sample can move a sample across an integer).  Coordinates sitting exactly on,
    one ulp below / above, and up to 2e-2 around integers — the band where the predicate switches — in both kernel variants
    (B = 1: 8-wave small form, B = 20: 4-queries-per-wave batched form at 60 x 80), against ATen grid_sample on the CPU:
    any cell the predicate wrongly skipped would show up as a missing O(16) contribution."""
    from macvo_amd import ops
    from oracle import corr

    H, W = (24, 40) if B == 1 else (60, 80)
    g = torch.Generator().manual_seed(77)
    vol = torch.randn(B * H * W, 1, H, W, generator=g) * 16
    base = corr.coords_grid(B, H, W)
    eps = torch.tensor([0.0, 1.2e-7, -1.2e-7, 1e-6, -1e-6, 1e-4, -1e-4, 5e-3, -5e-3, 9.9e-3, -9.9e-3, 1.01e-2, -1.01e-2, 2e-2, -2e-2, 0.5])
    pick = torch.randint(0, len(eps), (B, 2, H, W), generator=g)
    shift = torch.randint(-6, 7, (B, 2, H, W), generator=g).float()          # whole-pixel motion: windows cross the image border
    coords = base + shift + eps[pick]
    ref = corr.corr_lookup(vol, coords, 4)
    out = ops.corr_lookup(vol.to(gpu), coords.to(gpu), 4).cpu()
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=2e-4)
    # and bit-identical to the same lookup with every coordinate's margins forced on (fraction 0 -> all four flags cannot be set at
    # once, so compare against the property instead): shifting the volume content by one cell and the coordinates with it
    vol2 = torch.roll(vol, shifts=(1, 1), dims=(2, 3))
    out2 = ops.corr_lookup(vol2.to(gpu), (coords + 1.0).to(gpu), 4).cpu()
    ref2 = corr.corr_lookup(vol2, coords + 1.0, 4)
    torch.testing.assert_close(out2, ref2, rtol=1e-5, atol=2e-4)


def test_corr_lookup_far_outside_and_nan(gpu):
    """Coordinates far outside the slice give exact zeros (zero padding); NaN coordinates must not fault."""
    from macvo_amd import ops
    from oracle import corr

    B, H, W = 1, 8, 8
    vol = torch.ones(B * H * W, 1, H, W)
    coords = corr.coords_grid(B, H, W) + 1000.0
    out = ops.corr_lookup(vol.to(gpu), coords.to(gpu), 4).cpu()
    assert torch.count_nonzero(out) == 0
    coords[0, 0, 0, 0] = float("nan")
    ops.corr_lookup(vol.to(gpu), coords.to(gpu), 4)
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,h,w", [(2, 60, 80), (1, 7, 9), (3, 16, 24)])
def test_convex_upsample(gpu, B, h, w):
    """§8(f) rank 1: convex 8x upsampling vs the oracle (softmax over 9 taps of unfold(8*flow)); exp(2x) fused variant."""
    from macvo_amd import ops
    from oracle import frontend

    g = torch.Generator().manual_seed(31)
    flow8 = torch.randn(B, 2, h, w, generator=g) * 3
    mask = torch.randn(B, 576, h, w, generator=g) * 4
    ref = frontend.upsample_flow(flow8, 0.25 * mask)
    out = ops.convex_upsample(flow8.to(gpu), mask.to(gpu), mask_scale=0.25).cpu()
    assert out.shape == (B, 2, 8 * h, 8 * w)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=2e-5)
    cov8 = torch.randn(B, 2, h, w, generator=g) * 0.05
    ref2 = torch.exp(frontend.upsample_flow(cov8, mask) * 2)
    out2 = ops.convex_upsample(cov8.to(gpu), mask.to(gpu), mask_scale=1.0, exp2_out=True).cpu()
    torch.testing.assert_close(out2, ref2, rtol=2e-5, atol=1e-6)
    # convexity: a constant field is reproduced exactly in the interior (weights sum to 1)
    const = torch.full((1, 2, h, w), 0.5)
    up = ops.convex_upsample(const.to(gpu), mask[:1].to(gpu)).cpu()
    assert (up[..., 8:-8, 8:-8] - 4.0).abs().max() < 1e-5


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,h,w", [(2, 60, 80), (1, 7, 9), (1, 90, 160), (3, 16, 24), (1, 1, 1), (1, 5, 13)])
def test_convex_upsample_16bit_mask(gpu, dtype, B, h, w):
    """Fast mode (MACVO_Fast.yaml:73-74): the mask arrives in the decoder's autocast type and is read as it is — the result must be the
    fp32 kernel's on the widened mask, bit for bit (same arithmetic on the same fp32 values), for even and odd plane sizes (dword pairs /
    16-bit loads) and through the half-width launch of small frames."""
    from macvo_amd import ops

    g = torch.Generator().manual_seed(32)
    flow8 = (torch.randn(B, 2, h, w, generator=g) * 3).to(gpu)
    mask = (torch.randn(B, 576, h, w, generator=g) * 4).to(dtype).to(gpu)
    for scale, e2 in ((0.25, False), (1.0, True)):
        fl = flow8 * (0.02 if e2 else 1.0)
        a = ops.convex_upsample(fl, mask, mask_scale=scale, exp2_out=e2)
        b = ops.convex_upsample(fl, mask.float(), mask_scale=scale, exp2_out=e2)
        assert torch.equal(a, b)
    # an odd element offset into the storage (2-byte aligned only): the pair path must not be taken blindly
    buf = torch.zeros(mask.numel() + 1, dtype=dtype, device=gpu)
    buf[1:] = mask.flatten()
    assert torch.equal(ops.convex_upsample(flow8, buf[1:].view_as(mask), 0.25), ops.convex_upsample(flow8, mask.float(), 0.25))


def test_convex_upsample_minus_inf_taps(gpu):
    """an fp16 mask head that overflows under autocast hands the kernel -inf logits.  torch.softmax gives such a tap weight 0 and the output stays
    finite; the kernel's short exp must do the same (a NaN logit still poisons its sub-pixel, as in torch)."""
    from macvo_amd import ops
    from oracle import frontend

    g = torch.Generator().manual_seed(33)
    B, h, w = 1, 12, 16
    flow8 = torch.randn(B, 2, h, w, generator=g) * 3
    mask = torch.randn(B, 576, h, w, generator=g) * 4
    mask.view(B, 9, 64, h, w)[:, 3, ::2] = float("-inf")        # tap 3 of every second sub-pixel
    mask.view(B, 9, 64, h, w)[:, 7, 5, 2:5] = -3.0e38            # ... and logits whose log2(e) multiple overflows
    for dt in (torch.float32, torch.float16):
        m = mask.to(dt)
        ref = frontend.upsample_flow(flow8, m.float())
        out = ops.convex_upsample(flow8.to(gpu), m.to(gpu), mask_scale=1.0).cpu()
        assert torch.isfinite(ref).all() and torch.isfinite(out).all()
        torch.testing.assert_close(out, ref, rtol=1e-5, atol=2e-5)
    m = mask.clone()
    m.view(B, 9, 64, h, w)[0, 0, 0, 0, 0] = float("nan")
    out = ops.convex_upsample(flow8.to(gpu), m.to(gpu), mask_scale=1.0).cpu()
    assert torch.isnan(out[0, :, 0, 0]).all() and torch.isfinite(out[0, :, 8:, 8:]).all()

@pytest.mark.parametrize("shape", [(1, 32, 112, 160), (2, 196, 7, 10), (1, 128, 14, 20), (2, 96, 28, 40), (1, 64, 56, 80),
                                   (1, 16, 33, 47), (1, 3, 5, 5), (3, 8, 1, 1)])
def test_local_corr81(gpu, shape):
    """PWC-Net 81-channel local correlation (§8(f) rank 3) at the pyramid shapes of pwc_model.py:178-233 for a 640x448
    input, plus ragged / degenerate sizes.  Oracle = the definition in float64; tolerance = fp32 accumulation over C
    (1e-6 relative to (1/C) * sum |a||b|)."""
    from macvo_amd import ops
    from oracle import corr

    B, C, H, W = shape
    g = torch.Generator().manual_seed(C + H)
    a, b = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
    ref = corr.local_corr81(a, b, torch.float64)
    out = ops.local_corr81(a.to(gpu), b.to(gpu)).cpu()
    assert out.shape == (B, 81, H, W)
    scale = corr.local_corr81(a.abs(), b.abs(), torch.float64)
    err = (out.double() - ref).abs()
    assert (err <= 1e-6 * scale + 1e-30).all(), float((err / scale.clamp_min(1e-30)).max())
    # zero padding: displacement (-4,-4) at the top-left pixel reads outside the image
    assert out[:, 0, 0, 0].abs().max() == 0


def test_function_correlation_drop_in(gpu):
    from macvo_amd import plugins
    from oracle import corr

    a, b = torch.randn(1, 32, 20, 24), torch.randn(1, 32, 20, 24)
    with torch.no_grad():
        out = plugins.FunctionCorrelation(tenFirst=a.to(gpu), tenSecond=b.to(gpu))
    torch.testing.assert_close(out.cpu(), corr.local_corr81(a, b), rtol=1e-4, atol=1e-5)
    with pytest.raises(NotImplementedError):
        plugins.FunctionCorrelation(tenFirst=a, tenSecond=b)        # CPU: same behaviour as the reference
    with pytest.raises(AssertionError):
        plugins.FunctionCorrelation(tenFirst=a.to(gpu).transpose(2, 3), tenSecond=b.to(gpu))
