"""CPU restatement of FlowFormer's cost PATCH EMBEDDING ``proj`` stack (TEST INFRASTRUCTURE — nothing under mac-vo_amd/ imports this).

Path: ``Module/Network/FlowFormerCov/flownet.py:26`` calls ``memory_encoder(image1, image2, ...)``; FlowFormer's ``MemoryEncoder`` turns every
``H2 x W2`` slice of the all-pairs volume into ``H2/8 x W2/8`` cost tokens with ``PatchEmbed(patch_size = 8, in_chans = cost_heads_num = 1,
embed_dim = cost_latent_input_dim = 64)`` (``Config/Train/Demo.yaml:20-36``; the token count is what ``covhead.py:61-64`` documents as
``cost_memory``'s ``H2'·W2'``).  The FlowFormer submodule (MAC-VO/S_FlowFormer) is EMPTY in the reference checkout, so this follows the
published FlowFormer sources (core/FlowFormer/LatentCostFormer/encoder.py, ``PatchEmbed.__init__`` / ``.forward``):

    pad_r = (8 - W % 8) % 8;  pad_b = (8 - H % 8) % 8;  x = F.pad(x, (0, pad_r, 0, pad_b))
    proj = Conv2d(1, 16, 6, stride 2, padding 2) -> ReLU -> Conv2d(16, 32, 6, 2, 2) -> ReLU -> Conv2d(32, 64, 6, 2, 2)

**parity unpinned** against the MAC-VO fork (no source, no weights); pinned to torch's own ``F.conv2d`` on the same weights.  What follows the
stack in ``PatchEmbed.forward`` (position encoding, two 1x1 convolutions, LayerNorm) stays PyTorch and is not restated here.

``patch_embed_proj_bf16 / _f16`` additionally round the slice, the weights and the two intermediate maps to bfloat16 / float16 — the arithmetic
of the HIP kernel for either operand type (16-bit operands, fp32 accumulation), so that the kernel can be held to a tight tolerance and the
16-bit error itself to a separate, looser one."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def make_weights(seed: int = 0, scale: float = 1.0):
    """Conv2d-default-like weights (uniform +-1/sqrt(fan_in)) for the three layers: (w1, b1, w2, b2, w3, b3), fp32."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for cout, cin in ((16, 1), (32, 16), (64, 32)):
        bound = scale / (cin * 36) ** 0.5
        out.append((torch.rand(cout, cin, 6, 6, generator=g) * 2 - 1) * bound)
        out.append((torch.rand(cout, generator=g) * 2 - 1) * bound)
    return tuple(out)


def _pad8(x: torch.Tensor) -> torch.Tensor:
    H, W = x.shape[-2:]
    return F.pad(x, (0, (8 - W % 8) % 8, 0, (8 - H % 8) % 8))


def patch_embed_proj(cost_maps: torch.Tensor, w1, b1, w2, b2, w3, b3) -> torch.Tensor:
    """cost_maps [S, 1, H2, W2] fp32 -> [S, 64, ceil(H2/8), ceil(W2/8)] fp32 (PatchEmbed.proj after the pad)."""
    x = _pad8(cost_maps.float())
    x = F.relu(F.conv2d(x, w1, b1, stride=2, padding=2))
    x = F.relu(F.conv2d(x, w2, b2, stride=2, padding=2))
    return F.conv2d(x, w3, b3, stride=2, padding=2)


def patch_embed_proj_16(cost_maps: torch.Tensor, w1, b1, w2, b2, w3, b3, dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """The same stack with 16-bit operands (slice, weights, intermediate maps rounded to ``dtype`` = bfloat16 or float16) and fp32
    accumulation / biases — the arithmetic of the HIP kernel for ``operand`` "bf16" / "f16"."""
    r = lambda t: t.to(dtype).float()   # noqa: E731
    x = r(_pad8(cost_maps.float()))
    x = r(F.relu(F.conv2d(x, r(w1), b1, stride=2, padding=2)))
    x = r(F.relu(F.conv2d(x, r(w2), b2, stride=2, padding=2)))
    return F.conv2d(x, r(w3), b3, stride=2, padding=2)


def patch_embed_proj_bf16(cost_maps: torch.Tensor, w1, b1, w2, b2, w3, b3) -> torch.Tensor:
    return patch_embed_proj_16(cost_maps, w1, b1, w2, b2, w3, b3, torch.bfloat16)


def patch_embed_proj_f16(cost_maps: torch.Tensor, w1, b1, w2, b2, w3, b3) -> torch.Tensor:
    return patch_embed_proj_16(cost_maps, w1, b1, w2, b2, w3, b3, torch.float16)


def to_tokens(y: torch.Tensor) -> torch.Tensor:
    """[S, C, h, w] -> [S, h*w, C] (``x.flatten(2).transpose(1, 2)``, the layout the encoder's token layers consume)"""
    return y.flatten(2).transpose(1, 2).contiguous()
