"""GPU parity of the split + streaming cost volume (csrc/corr_volume_split.hip): fp32 feature maps packed into three bf16
pieces, six piece products on the 16-bit matrix pipe, fp32 accumulate.  The bar is the EXACT fp32 path's
(|out - einsum_f64| <= 2e-5 sqrt(C), tests/test_gpu_corr.py::test_corr_volume_f32_chw) on every shape of that test the kernel
covers, plus 1280x720, B = 64, ragged row counts, wide dynamic range, and size-independent properties."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _feats(B, C, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)


def test_pack_layout_and_piece_sums(gpu):
    """mv_volume_pack: unit (b, rb, ks, piece) holds, for lane = kh * 32 + li, the 8 values k = 16 ks + 8 kh + e of row 32 rb + li;
    pieces are round-to-nearest bf16 of the running residual (so p0 + p1 + p2 reproduces x to 2^-24 |x| and p0 = bf16(x)); rows
    past N and the extra row block repeat row N - 1.  CHW and HWC inputs give the same bytes."""
    from macvo_amd import ops

    B, C, N1, N2 = 2, 256, 100, 128
    g = torch.Generator().manual_seed(0)
    f1, f2 = torch.randn(B, N1, C, generator=g), torch.randn(B, N2, C, generator=g)
    f1[0, 3] *= 1e4
    f1[1, 7] *= 1e-4
    p1, p2 = ops.volume_pack(f1.to(gpu), f2.to(gpu), layout="hwc")
    q1, q2 = ops.volume_pack(f1.permute(0, 2, 1).contiguous().to(gpu).view(B, C, 10, 10), f2.permute(0, 2, 1).contiguous().to(gpu).view(B, C, 8, 16), layout="chw")
    assert torch.equal(p1, q1) and torch.equal(p2, q2)
    for f, p, N in ((f1, p1, N1), (f2, p2, N2)):
        nrb = (N + 31) // 32 + 1
        u = p.cpu().view(torch.bfloat16).view(B, nrb, C // 16, 3, 2, 32, 8).float()        # [b, rb, ks, piece, kh, li, e]
        rows = torch.arange(nrb * 32).clamp_max(N - 1)
        rows[(nrb - 1) * 32:] = N - 1
        want = f[:, rows].view(B, nrb, 32, C // 16, 2, 8).permute(0, 1, 3, 4, 2, 5)          # [b, rb, ks, kh, li, e]
        assert torch.equal(u[:, :, :, 0], want.to(torch.bfloat16).float())                    # leading piece = bf16(x)
        s = (u[:, :, :, 0].double() + u[:, :, :, 1].double() + u[:, :, :, 2].double())
        assert ((s - want.double()).abs() <= want.double().abs() * 2.0 ** -23).all()
        r1 = want - u[:, :, :, 0]
        assert torch.equal(u[:, :, :, 1], r1.to(torch.bfloat16).float())                      # second piece = bf16(residual)


def test_pack_f16x2_row_scales_and_pieces(gpu):
    """mode "f16x2": every row is scaled by 2^sh with its largest magnitude landing in [2^14, 2^15) (sh in the int32 table behind
    the units, 0 for all-zero rows), pieces are fp16(x 2^sh) and fp16 of the exact residual: (p0 + p1) 2^-sh reproduces x to
    2^-21 |x| + 2^-40 max|row| — whatever the magnitude of the row."""
    from macvo_amd import ops

    B, C, N = 2, 256, 96
    g = torch.Generator().manual_seed(1)
    f = torch.randn(B, N, C, generator=g)
    f[0, 3] *= 1e6
    f[0, 4] *= 1e-6
    f[1, 5] = 0.0
    f[1, 6, :200] *= 1e-5                       # wide range INSIDE a row: small entries keep absolute, not relative, accuracy
    p, _ = ops.volume_pack(f.to(gpu), f.to(gpu), layout="hwc", mode="f16x2")
    nrb = N // 32 + 1
    units = B * nrb * (C // 16) * 2 * 1024
    u = p[:units].cpu().view(torch.float16).view(B, nrb, C // 16, 2, 2, 32, 8).double()      # [b, rb, ks, piece, kh, li, e]
    sh = p[units: units + B * nrb * 32 * 4].cpu().view(torch.int32).view(B, nrb, 32)
    rows = torch.arange(nrb * 32).clamp_max(N - 1)
    rows[(nrb - 1) * 32:] = N - 1
    want = f[:, rows].view(B, nrb, 32, C // 16, 2, 8).permute(0, 1, 3, 4, 2, 5).double()       # [b, rb, ks, kh, li, e]
    rmax = f[:, rows].abs().amax(-1).view(B, nrb, 32)
    scaled_max = rmax.double() * torch.pow(2.0, sh.double())
    nz = rmax > 0
    assert ((scaled_max[nz] >= 2.0 ** 14) & (scaled_max[nz] < 2.0 ** 15)).all() and (sh[~nz] == 0).all()
    back = (u[:, :, :, 0] + u[:, :, :, 1]) * torch.pow(2.0, -sh.double())[:, :, None, None, :, None]
    tol = want.abs() * 2.0 ** -21 + rmax.double()[:, :, None, None, :, None] * 2.0 ** -39
    assert ((back - want).abs() <= tol).all()


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("shape", [(2, 256, 16, 24), (1, 256, 60, 80), (2, 256, 60, 80), (3, 256, 59, 64), (1, 256, 8, 8)])
@pytest.mark.parametrize("layout", ["chw", "hwc"])
def test_corr_volume_split_parity(gpu, shape, layout, mode):
    from macvo_amd import ops
    from oracle import corr

    B, C, H, W = shape
    f1, f2 = _feats(B, C, H, W, seed=0)
    ref64 = corr.corr_volume(f1, f2, torch.float64)
    a1, a2 = (f1, f2) if layout == "chw" else (f1.permute(0, 2, 3, 1).contiguous(), f2.permute(0, 2, 3, 1).contiguous())
    out = ops.corr_volume(a1.to(gpu), a2.to(gpu), layout=layout, precision=mode)
    assert ops.last_volume_kernel() == f"corr_volume_split_stream<{mode}>"
    out = out.cpu()
    assert out.shape == (B * H * W, 1, H, W) and out.dtype == torch.float32
    err = (out.double() - ref64).abs().max().item()
    assert err <= 2e-5 * float(C) ** 0.5, err
    ref32 = corr.corr_volume(f1, f2, torch.float32)
    assert err <= 4 * (ref32.double() - ref64).abs().max().item() + 1e-6      # and not worse than a float32 einsum on the CPU
    exact = ops.corr_volume(a1.to(gpu), a2.to(gpu), layout=layout, precision="exact").cpu()
    assert err <= 1.5 * (exact.double() - ref64).abs().max().item() + 1e-6     # ... nor than the exact fp32 MFMA path


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2"])
def test_corr_volume_split_ragged_rows_and_dynamic_range(gpu, mode):
    """N1 not a multiple of 32 / 128 (row replication, replica block, waves past the bottom edge), N1 != N2, and rows scaled by
    1e3 / 1e-3 (f16x2: up to 2^+-100 — the per-row power-of-two scales carry them and are not clamped: no row of the fp32 range can
    overflow the fp16 pieces): the error relative to sum |a||b| stays at the fp32 level."""
    from macvo_amd import ops

    C = 256
    g = torch.Generator().manual_seed(7)
    for B, N1, N2 in ((2, 100, 128), (1, 4800 - 17, 4800), (2, 33, 64), (1, 129, 192)):
        f1, f2 = torch.randn(B, N1, C, generator=g), torch.randn(B, N2, C, generator=g)
        f1[0, 0] *= 1e3
        f2[0, 1] *= 1e-3
        f1[0, N1 - 1] *= 17.0
        if mode == "f16x2":
            f1[0, 2] *= 1e12
            f2[0, 3] *= 1e-12
            f1[0, 5] = 0.0
            f1[0, 7] *= 2.0 ** 100                 # outside the round-3 clamp (rows above 2^74 overflowed the fp16 pieces to inf)
            f2[0, 9] *= 2.0 ** -100                # (2^100 x 2^-100 and 2^100 x O(1e3) stay finite in fp32)
            f1[0, 11] *= 2.0 ** -20                # (every pairwise product stays inside fp32's normal range: 2^-120 .. 2^115)
            f2[0, 13] *= 2.0 ** 10
        out = ops.corr_volume(f1.to(gpu), f2.to(gpu), layout="hwc", precision=mode)
        assert ops.last_volume_kernel() == f"corr_volume_split_stream<{mode}>"
        out = out.cpu().view(B, N1, N2).double()
        ref = torch.einsum("bid,bjd->bij", f1.double(), f2.double())
        scale = torch.einsum("bid,bjd->bij", f1.double().abs(), f2.double().abs())
        assert torch.isfinite(out).all()
        assert ((out - ref).abs() / scale.clamp_min(1e-300)).max().item() <= 1e-6, (B, N1, N2)
        assert (out[0, 5] == 0).all() if mode == "f16x2" else True


def test_corr_volume_bf16x3_falls_back_to_exact_outside_its_shapes(gpu):
    from macvo_amd import ops
    from oracle import corr

    f1, f2 = _feats(1, 64, 8, 12, seed=1)                    # C = 64: not covered by the streaming kernel
    out = ops.corr_volume(f1.to(gpu), f2.to(gpu), precision="bf16x3")
    assert not ops.last_volume_kernel().startswith("corr_volume_split")
    assert torch.equal(out, ops.corr_volume(f1.to(gpu), f2.to(gpu)))
    assert (out.cpu().double() - corr.corr_volume(f1, f2, torch.float64)).abs().max() <= 2e-5 * 8


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("B,H,W", [(2, 90, 160), (64, 60, 80)])
def test_corr_volume_split_fullsize_sampled_rows_and_properties(gpu, B, H, W, mode):
    """BASELINE configs[2] (1280x720, N = 14400) and configs[4] (B = 64 pairs): sampled query rows + edge rows vs fp64, exact
    homogeneity under a power-of-two scale (pieces scale exactly), batch independence (bitwise)."""
    from macvo_amd import ops

    C = 256
    f1, f2 = _feats(B, C, H, W, seed=0)
    N = H * W
    d1, d2 = f1.to(gpu), f2.to(gpu)
    vol = ops.corr_volume(d1, d2, precision=mode)
    assert ops.last_volume_kernel() == f"corr_volume_split_stream<{mode}>" and vol.shape == (B * N, 1, H, W)
    a, b_ = f1.reshape(B, C, N).double(), f2.reshape(B, C, N).double()
    idx = torch.cat([torch.arange(0, B * N, 1009 if B == 2 else 30011), torch.tensor([0, N - 1, B * N - 1, (B - 1) * N, B * N - 64, B * N - 65])])
    ref = torch.stack([a[int(i) // N, :, int(i) % N] @ b_[int(i) // N] for i in idx])
    got = vol[idx.to(gpu)].reshape(len(idx), -1).cpu().double()
    assert (got - ref).abs().max().item() <= 2e-5 * C ** 0.5
    chk = vol[:: 997].clone()
    vol2 = ops.corr_volume(d1 * 4.0, d2, precision=mode)
    assert torch.equal(vol2[:: 997], chk * 4.0)
    del vol2
    b = B - 1
    solo = ops.corr_volume(d1[b:b + 1].contiguous(), d2[b:b + 1].contiguous(), precision=mode)
    assert torch.equal(solo, vol[b * N:(b + 1) * N])
    del solo
    # ... and EVERY cell against the exact fp32 MFMA kernel (the hand-placed waits / the M0 contract of the LDS-DMA groups are invisible to hipcc; a
    # corrupted fragment would show up in some band of some pair, not necessarily in the sampled rows): both are within 2e-5 sqrt(C) of the fp64 product
    exact = ops.corr_volume(d1, d2, precision="exact")
    worst = 0.0
    for r0 in range(0, B * N, 4 * N):                       # chunked: the difference of two 5.9-GB volumes is not materialised at once
        worst = max(worst, float((vol[r0:r0 + 4 * N] - exact[r0:r0 + 4 * N]).abs().max()))
    assert worst <= 4e-5 * C ** 0.5, worst


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2"])
def test_corr_volume_split_is_deterministic_and_item_order_free(gpu, mode):
    """Two launches give the same bits, and so do different XCD region counts (MV_SPLIT_REGIONS only reorders whole items)."""
    import hashlib, os, subprocess, sys

    code = r'''
import hashlib, sys, torch
sys.path.insert(0, sys.argv[1])
from macvo_amd import ops
g = torch.Generator().manual_seed(3)
f1 = torch.randn(2, 256, 60, 80, generator=g).cuda(); f2 = torch.randn(2, 256, 60, 80, generator=g).cuda()
a = ops.corr_volume(f1, f2, precision=sys.argv[2]); b = ops.corr_volume(f1, f2, precision=sys.argv[2])
assert torch.equal(a, b)
print(hashlib.sha1(a.cpu().numpy().tobytes()).hexdigest())
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shas = []
    for regions in ("1", "4", "7"):
        r = subprocess.run([sys.executable, "-c", code, root, mode], env=dict(os.environ, MV_SPLIT_REGIONS=regions), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        shas.append(r.stdout.split()[-1])
    assert len(set(shas)) == 1, shas


@pytest.mark.parametrize("mode", ["f16x2", "bf16x3"])
@pytest.mark.parametrize("B,H,W", [(2, 60, 80), (6, 24, 32), (1, 8, 8)])
def test_tiled_volume_and_tiled_lookup_equal_the_row_major_forms(gpu, mode, B, H, W):
    """`mv_volume_pack_tiled` permutes operand 2's pixels into 4 x 4-tile order, so the unchanged GEMM writes every query's slice
    tiled; `mv_corr_lookup_tiled` reads that layout.  (i) the tiled volume is the row-major one permuted, bit for bit (each output
    element is the same k-ordered sum wherever its column sits); (ii) the tiled lookup returns the row-major lookup's tokens bit for
    bit — for coordinates inside, across the border, far outside, on / next to integers (the margin-tile predicate) and in both
    kernel variants (B * N <= / > the small-launch threshold)."""
    from macvo_amd import ops
    from oracle import corr

    C = 256
    N = H * W
    f1, f2 = _feats(B, C, H, W, seed=5)
    d1, d2 = f1.to(gpu), f2.to(gpu)
    pk = ops.volume_pack(d1, d2, mode=mode)
    vol = ops.corr_volume_packed(pk[0], pk[1], B, C, N, N, mode=mode).view(B * N, 1, H, W)
    pkt = ops.volume_pack(d1, d2, mode=mode, tiled_hw=(H, W))
    volt = ops.corr_volume_packed(pkt[0], pkt[1], B, C, N, N, mode=mode).view(B * N, 1, H, W)
    assert ops.last_volume_kernel() == f"corr_volume_split_stream<{mode}>"
    # (i) un-tile on the host: [q][ty][tx][4][4] -> [q][y][x]
    unt = volt.view(B * N, H // 4, W // 4, 4, 4).permute(0, 1, 3, 2, 4).reshape(B * N, 1, H, W)
    assert torch.equal(unt, vol)
    # (ii) lookups
    g = torch.Generator().manual_seed(9)
    eps = torch.tensor([0.0, 1.2e-7, -1.2e-7, 1e-4, -1e-4, 9.9e-3, -9.9e-3, 1.01e-2, -1.01e-2, 0.3, 0.5, -0.4])
    for it in range(3):
        pick = torch.randint(0, len(eps), (B, 2, H, W), generator=g)
        shift = torch.randint(-7, 8, (B, 2, H, W), generator=g).float()
        coords = corr.coords_grid(B, H, W) + shift + eps[pick] + (torch.rand(B, 2, H, W, generator=g) * 4 - 2) * (it == 2)
        if it == 1:
            coords[:, :, 0, 0] = 1000.0                                   # far outside: zeros
            coords[:, 0, 1, 1] = -3.0                                     # window across the left border
        cd = coords.to(gpu)
        a = ops.corr_lookup(vol, cd, 4)
        b_ = ops.corr_lookup(volt, cd, 4, tiled=True)
        assert torch.equal(a, b_), it
    for thr in ("0", "1000000000"):                                       # force the other kernel variant through its A/B knob
        import os, subprocess, sys
        code = (
            "import sys, torch; sys.path.insert(0, sys.argv[1])\n"
            "from macvo_amd import ops\n"
            "from oracle import corr\n"
            f"B, C, H, W = {B}, 256, {H}, {W}; N = H * W\n"
            "g = torch.Generator().manual_seed(5)\n"
            "f1 = torch.randn(B, C, H, W, generator=g).cuda(); f2 = torch.randn(B, C, H, W, generator=g).cuda()\n"
            f"pk = ops.volume_pack(f1, f2, mode='{mode}'); pkt = ops.volume_pack(f1, f2, mode='{mode}', tiled_hw=(H, W))\n"
            f"vol = ops.corr_volume_packed(pk[0], pk[1], B, C, N, N, mode='{mode}').view(B * N, 1, H, W)\n"
            f"volt = ops.corr_volume_packed(pkt[0], pkt[1], B, C, N, N, mode='{mode}').view(B * N, 1, H, W)\n"
            "coords = (corr.coords_grid(B, H, W) + torch.rand(B, 2, H, W, generator=g) * 12 - 6).cuda()\n"
            "assert torch.equal(ops.corr_lookup(vol, coords, 4), ops.corr_lookup(volt, coords, 4, tiled=True))\n"
        )
        if (B, H, W) != (6, 24, 32):
            continue
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        r = subprocess.run([sys.executable, "-c", code, root], env=dict(os.environ, MV_LOOKUP_SMALL=thr), capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
