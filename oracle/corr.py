"""All-pairs cost volume + 9x9 window lookup of FlowFormer (TEST INFRASTRUCTURE).

The arithmetic lives in the reference's *empty, un-pinned* git submodule
``Module/Network/FlowFormer`` (``.gitmodules:1-3`` -> MAC-VO/S_FlowFormer); the only
in-tree evidence is the call sites ``Module/Network/FlowFormerCov/flownet.py:26-27``
(``memory_encoder`` returns ``cost_maps``; cast to fp32) and
``Module/Network/FlowFormerCov/covhead.py:91-92`` (``encode_flow_token(cost_maps, coords1)``,
"MUST run in fp32"), with hyper-parameters at ``Config/Train/Demo.yaml:20-61``
(cost_heads_num 1, encoder_latent_dim 256, kernel_size 9).  The functions below restate
the published FlowFormer / RAFT algorithm (LatentCostFormer ``MemoryEncoder.corr`` and
``MemoryDecoder.encode_flow_token`` / RAFT ``CorrBlock`` + ``bilinear_sampler``):
"parity unpinned" w.r.t. the MAC-VO fork, pinned to ``einsum`` / ``grid_sample`` semantics.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def corr_volume(f1: torch.Tensor, f2: torch.Tensor, accum_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """``corr = einsum('bhid,bhjd->bhij')`` with heads=1 and NO 1/sqrt(d) scaling, reshaped to
    ``cost_maps [B*H1*W1, 1, H2, W2]`` (one H2xW2 slice per source pixel).

    f1, f2: feature maps ``[B, C, H, W]`` (any float dtype).  Products are accumulated in
    ``accum_dtype`` (float32 = what the fp32 reference path does; float64 = error yardstick).
    """
    B, C, H1, W1 = f1.shape
    _, _, H2, W2 = f2.shape
    a = f1.reshape(B, C, H1 * W1).permute(0, 2, 1).to(accum_dtype)  # b (y x) d
    b = f2.reshape(B, C, H2 * W2).permute(0, 2, 1).to(accum_dtype)
    corr = torch.einsum("bid,bjd->bij", a, b)
    return corr.reshape(B * H1 * W1, 1, H2, W2)


def coords_grid(B: int, H: int, W: int, dtype=torch.float32) -> torch.Tensor:
    """``initialize_flow`` / RAFT ``coords_grid``: channel 0 = x, channel 1 = y (in-tree twin:
    ``Module/Network/PWCNet/pwc_cov/gru.py:8-21``)."""
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    return torch.stack([xs, ys], dim=0).to(dtype)[None].repeat(B, 1, 1, 1)


def bilinear_sampler(img: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """RAFT ``bilinear_sampler``: pixel coords -> [-1, 1] by (W-1), (H-1); grid_sample with
    align_corners=True and zero padding."""
    H, W = img.shape[-2:]
    xgrid, ygrid = coords.split([1, 1], dim=-1)
    xgrid = 2 * xgrid / (W - 1) - 1
    ygrid = 2 * ygrid / (H - 1) - 1
    grid = torch.cat([xgrid, ygrid], dim=-1)
    return F.grid_sample(img, grid, align_corners=True)


def corr_lookup(cost_maps: torch.Tensor, coords: torch.Tensor, radius: int = 4) -> torch.Tensor:
    """``MemoryDecoder.encode_flow_token(cost_maps, coords)``.

    cost_maps ``[B*H1*W1, 1, H2, W2]`` fp32, coords ``[B, 2, H1, W1]`` fp32 (x, y) ->
    ``[B, (2r+1)^2, H1, W1]``; output channel ``k = (2r+1)*i + j`` samples the query's own
    slice at ``(x + (i - r), y + (j - r))`` (RAFT's dy/dx stacking quirk: first window index
    moves along x).
    """
    coords = coords.permute(0, 2, 3, 1)
    batch, h1, w1, _ = coords.shape
    r = radius
    dx = torch.linspace(-r, r, 2 * r + 1)
    dy = torch.linspace(-r, r, 2 * r + 1)
    delta = torch.stack(torch.meshgrid(dy, dx, indexing="ij"), dim=-1).to(coords)
    centroid = coords.reshape(batch * h1 * w1, 1, 1, 2)
    delta = delta.view(1, 2 * r + 1, 2 * r + 1, 2)
    corr = bilinear_sampler(cost_maps, centroid + delta)
    return corr.view(batch, h1, w1, -1).permute(0, 3, 1, 2).contiguous()


def corr_lookup_naive(cost_maps: torch.Tensor, coords: torch.Tensor, radius: int = 4) -> torch.Tensor:
    """Independent restatement of :func:`corr_lookup` without ``grid_sample`` (explicit floor +
    4-tap bilinear with zero padding, same fp32 normalise/un-normalise round trip that
    ``grid_sample(align_corners=True)`` performs).  Used to cross-check the oracle itself."""
    BN, _, H2, W2 = cost_maps.shape
    B, _, H1, W1 = coords.shape
    r = radius
    K = 2 * r + 1
    vol = cost_maps.reshape(B, H1 * W1, H2, W2)
    x = coords[:, 0].reshape(B, H1 * W1)
    y = coords[:, 1].reshape(B, H1 * W1)
    out = torch.zeros(B, K * K, H1 * W1, dtype=torch.float32)
    q = torch.arange(H1 * W1)
    for i in range(K):
        for j in range(K):
            xs = x + float(i - r)
            ys = y + float(j - r)
            # RAFT normalise, then ATen grid_sampler_unnormalize(align_corners=True)
            xg = 2 * xs / (W2 - 1) - 1
            yg = 2 * ys / (H2 - 1) - 1
            ix = ((xg + 1) / 2) * (W2 - 1)
            iy = ((yg + 1) / 2) * (H2 - 1)
            x0 = ix.floor()
            y0 = iy.floor()
            wx1 = ix - x0
            wy1 = iy - y0
            acc = torch.zeros(B, H1 * W1, dtype=torch.float32)
            for (xx, yy, w) in (
                (x0, y0, (x0 + 1 - ix) * (y0 + 1 - iy)),
                (x0 + 1, y0, wx1 * (y0 + 1 - iy)),
                (x0, y0 + 1, (x0 + 1 - ix) * wy1),
                (x0 + 1, y0 + 1, wx1 * wy1),
            ):
                ok = (xx >= 0) & (xx <= W2 - 1) & (yy >= 0) & (yy <= H2 - 1)
                xi = xx.clamp(0, W2 - 1).long()
                yi = yy.clamp(0, H2 - 1).long()
                for b in range(B):
                    v = vol[b, q, yi[b], xi[b]]
                    acc[b] += torch.where(ok[b], v * w[b], torch.zeros_like(v))
            out[:, K * i + j] = acc
    return out.reshape(B, K * K, H1, W1)


def local_corr81(first: torch.Tensor, second: torch.Tensor, accum_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """PWC-Net local correlation, forward: ``_FunctionCorrelation.forward`` (Module/Network/PWCNet/pwc/correlation.py:277-325)
    = ``kernel_Correlation_rearrange`` (:8-33, zero padding by 4) + ``kernel_Correlation_updateOutput`` (:35-103):

        out[b, 9*(s2p+4) + (s2o+4), y, x] = sum_c first[b,c,y,x] * second[b,c,y+s2p,x+s2o] / C      (:73-74,:99-101)

    with ``s2o = top_channel % 9 - 4`` the x offset and ``s2p = top_channel / 9 - 4`` the y offset.  PARITY UNPINNED: the
    reference kernel is CUDA-only (cupy; the CPU branch raises NotImplementedError, :323-324) and its tests hold no
    vectors for it, so this restates the arithmetic definition; the reference's own summation order (32 strided partial
    sums, serial reduce by lane 0, :76-96) differs from any other order at fp32 rounding level only.
    """
    B, C, H, W = first.shape
    a = first.to(accum_dtype)
    b = torch.nn.functional.pad(second.to(accum_dtype), (4, 4, 4, 4))
    out = torch.empty((B, 81, H, W), dtype=accum_dtype)
    for k in range(81):
        s2o, s2p = k % 9 - 4, k // 9 - 4
        out[:, k] = (a * b[:, :, 4 + s2p: 4 + s2p + H, 4 + s2o: 4 + s2o + W]).sum(dim=1) / C
    return out


def local_corr81_naive(first: torch.Tensor, second: torch.Tensor) -> torch.Tensor:
    """Per-pixel loops in float64 written directly from the kernel's index arithmetic (:47-101); tiny inputs only."""
    B, C, H, W = first.shape
    out = torch.zeros((B, 81, H, W), dtype=torch.float64)
    f, s = first.double(), second.double()
    for b in range(B):
        for y in range(H):
            for x in range(W):
                for k in range(81):
                    x2, y2 = x + k % 9 - 4, y + k // 9 - 4
                    if 0 <= x2 < W and 0 <= y2 < H:
                        out[b, k, y, x] = (f[b, :, y, x] * s[b, :, y2, x2]).sum() / C
    return out
