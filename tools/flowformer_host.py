"""A FlowFormerCov-shaped host network in plain PyTorch (measurement plumbing, NOT part of the drop-in).

Why it exists: the network the reference runs (``Module/Network/FlowFormerCov/flownet.py:9-44``) derives from the ``S_FlowFormer``
submodule, which is absent from the reference checkout (``.gitmodules:1-3``), and the pretrained weights are release downloads.  The
hot-path kernels are therefore exercised through hooks on stand-ins; this file supplies a *whole* network of the published FlowFormer
architecture (Huang et al., ECCV 2022, "latentcostformer" variant, hyper-parameters of ``Config/Train/Demo.yaml:20-61``) so that

* ``plugins.install_flowformer_hooks`` can be run against the attribute layout the public code has (``memory_encoder.corr``,
  ``memory_encoder.cost_perceiver_encoder.patch_embed.proj``, ``memory_decoder.encode_flow_token / upsample_flow``), and a hooked
  forward can be compared with the unhooked one end to end (``tests/test_gpu_flowformer_host.py``);
* an end-to-end figure (images -> pose, network included) can be measured on the MI355X (``bench.py`` ``end_to_end`` leg).

What is restated from where:

* ``FlowFormerCov.forward / inference`` — the in-tree ``flownet.py:18-44``; ``MemoryCovDecoder.forward``, ``CovHead``,
  ``CovUpdateBlock`` — the in-tree ``covhead.py:8-140`` (same statement order, same dtype switches, same "MUST run in fp32" islands);
* everything else (two-stage Twins-SVT-L encoder, cost perceiver encoder, cross-attention decoder layer, GMA update block,
  ``InputPadder``, ``initialize_flow``) — the PUBLIC FlowFormer / timm-Twins / GMA / RAFT architectures, from memory.  Layer shapes,
  tensor layouts and the kernel mix follow those publications; bit-level arithmetic of the absent submodule cannot be checked.
  **Parity unpinned** — weights are random, nothing here is a golden reference for anything.

The three methods the HIP library replaces are written here exactly as their published definitions (einsum / ``grid_sample`` /
softmax-unfold), and ``tests/test_flowformer_host.py`` pins them to ``oracle/corr.py`` — so the unhooked network is a valid "before"
for the hooked one.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


def demo_cfg(**over) -> SimpleNamespace:
    """``Config/Train/Demo.yaml:20-61`` (``latentcostformer`` block), the configuration every MAC-VO experiment file uses."""
    c = SimpleNamespace(pe="linear", dropout=0.0, encoder_latent_dim=256, query_latent_dim=64, cost_latent_input_dim=64,
                        cost_latent_token_num=8, cost_latent_dim=128, cost_heads_num=1, encoder_depth=3, patch_size=8,
                        kernel_size=9, vert_c_dim=64, cost_encoder_res=True, cnet="twins", fnet="twins", add_flow_token=True,
                        gma="GMA", decoder_depth=12)
    for k, v in over.items():
        setattr(c, k, v)
    return c


# ----------------------------------------------------------------------------------------------------------- small pieces
def coords_grid(batch: int, ht: int, wd: int, device=None) -> torch.Tensor:
    ys, xs = torch.meshgrid(torch.arange(ht, device=device), torch.arange(wd, device=device), indexing="ij")
    return torch.stack([xs, ys], dim=0).float()[None].repeat(batch, 1, 1, 1)


def sine_embedding(x: torch.Tensor, dim: int, scale: float = 1.0 / 200) -> torch.Tensor:
    """FlowFormer's "linear" position encoding of pixel coordinates ``x [..., 2]`` -> ``[..., dim]`` (sin/cos of x and y over dim/4 bands)."""
    bands = torch.linspace(0, dim // 4 - 1, dim // 4, device=x.device, dtype=x.dtype)
    ax, ay = 3.14 * x[..., 0:1] * bands * scale, 3.14 * x[..., 1:2] * bands * scale
    return torch.cat([ax.sin(), ax.cos(), ay.sin(), ay.cos()], dim=-1)


class InputPadder:
    """Replicate-pad to a multiple of 8, centred ("sintel" mode of the public utility the reference imports, flownet.py:4,37-43)."""

    def __init__(self, shape):
        ht, wd = shape[-2:]
        ph, pw = (-ht) % 8, (-wd) % 8
        self._pad = [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]

    def pad(self, *xs):
        return [F.pad(x, self._pad, mode="replicate") if any(self._pad) else x for x in xs]

    def unpad(self, x):
        ht, wd = x.shape[-2:]
        return x[..., self._pad[2]:ht - self._pad[3], self._pad[0]:wd - self._pad[1]]


def _mha(q, k, v, heads: int):
    """softmax(q k^T / sqrt(d)) v over ``[B, N, heads * d]`` tensors."""
    B, Nq, C = q.shape
    split = lambda t: t.view(t.shape[0], t.shape[1], heads, C // heads).transpose(1, 2)   # noqa: E731
    out = F.scaled_dot_product_attention(split(q), split(k), split(v))
    return out.transpose(1, 2).reshape(B, Nq, C)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(dim, hidden), nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


# ----------------------------------------------------------------------------------------------------------- Twins-SVT (2 stages)
def _to_windows(x, ws):
    """``[B, H, W, C]`` -> ``([B * nh * nw, ws * ws, C], (Hp, Wp))``, zero-padded on the right / bottom."""
    B, H, W, C = x.shape
    pr, pb = (-W) % ws, (-H) % ws
    if pr or pb:
        x = F.pad(x, (0, 0, 0, pr, 0, pb))
    Hp, Wp = H + pb, W + pr
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C).transpose(2, 3)
    return x.reshape(-1, ws * ws, C), (Hp, Wp)


def _from_windows(x, B, H, W, Hp, Wp, ws):
    C = x.shape[-1]
    x = x.view(B, Hp // ws, Wp // ws, ws, ws, C).transpose(2, 3).reshape(B, Hp, Wp, C)
    return x[:, :H, :W].reshape(B, H * W, C)


class LocalAttn(nn.Module):
    """Locally-grouped self-attention (Twins "LSA"): attention inside ws x ws windows."""

    def __init__(self, dim, heads, ws):
        super().__init__()
        self.heads, self.ws = heads, ws
        self.qkv, self.proj = nn.Linear(dim, 3 * dim), nn.Linear(dim, dim)

    def forward(self, x, size, context=None):
        B, N, C = x.shape
        H, W = size
        xw, (Hp, Wp) = _to_windows(x.view(B, H, W, C), self.ws)
        q, k, v = self.qkv(xw).chunk(3, dim=-1)
        return self.proj(_from_windows(_mha(q, k, v, self.heads), B, H, W, Hp, Wp, self.ws))


class GlobalAttn(nn.Module):
    """Global sub-sampled attention (Twins "GSA"): keys / values from an sr x sr strided convolution of the map."""

    def __init__(self, dim, heads, sr):
        super().__init__()
        self.heads = heads
        self.q, self.kv, self.proj = nn.Linear(dim, dim), nn.Linear(dim, 2 * dim), nn.Linear(dim, dim)
        self.sr, self.norm = nn.Conv2d(dim, dim, sr, sr), nn.LayerNorm(dim)

    def forward(self, x, size, context=None):
        B, N, C = x.shape
        sub = self.sr(x.transpose(1, 2).reshape(B, C, *size)).flatten(2).transpose(1, 2)
        k, v = self.kv(self.norm(sub)).chunk(2, dim=-1)
        return self.proj(_mha(self.q(x), k, v, self.heads))


class TwinsBlock(nn.Module):
    def __init__(self, dim, attn):
        super().__init__()
        self.norm1, self.attn, self.norm2, self.mlp = nn.LayerNorm(dim), attn, nn.LayerNorm(dim), Mlp(dim, 4 * dim)

    def forward(self, x, size, context=None):
        x = x + self.attn(self.norm1(x), size, context)
        return x + self.mlp(self.norm2(x))


class TwinsStage(nn.Module):
    def __init__(self, cin, dim, patch, heads, sr, ws=7, depth=2):
        super().__init__()
        self.embed, self.embed_norm = nn.Conv2d(cin, dim, patch, patch), nn.LayerNorm(dim)
        self.blocks = nn.ModuleList(TwinsBlock(dim, LocalAttn(dim, heads, ws) if j % 2 == 0 else GlobalAttn(dim, heads, sr)) for j in range(depth))
        self.pos = nn.Conv2d(dim, dim, 3, 1, 1, groups=dim)                      # conditional position encoding after the first block

    def forward(self, x):
        x = self.embed(x)
        B, C, H, W = x.shape
        x = self.embed_norm(x.flatten(2).transpose(1, 2))
        for j, blk in enumerate(self.blocks):
            x = blk(x, (H, W))
            if j == 0:
                m = x.transpose(1, 2).reshape(B, C, H, W)
                x = (self.pos(m) + m).flatten(2).transpose(1, 2)
        return x.transpose(1, 2).reshape(B, C, H, W)


class TwinsSVTLargeStem(nn.Module):
    """The first two stages of timm's ``twins_svt_large`` (dims 128 / 256, heads 4 / 8, depths 2 / 2, sr 8 / 4, windows 7), which is all
    FlowFormer keeps of it: ``[B, 3, H, W] -> [B, 256, H/8, W/8]``."""

    def __init__(self):
        super().__init__()
        self.stages = nn.ModuleList([TwinsStage(3, 128, 4, 4, 8), TwinsStage(128, 256, 2, 8, 4)])

    def forward(self, x):
        for s in self.stages:
            x = s(x)
        return x


# ----------------------------------------------------------------------------------------------------------- cost encoder
class PatchEmbed(nn.Module):
    """FlowFormer's cost-map patch embedding for ``patch_size`` 8: three 6x6 stride-2 convolutions (the (f)2 kernel replaces ``proj``),
    then a 1x1 FFN over [features, position code] and a LayerNorm: ``[S, heads, H2, W2] -> [S, H3 * W3, 2 * embed_dim]``."""

    def __init__(self, in_chans=1, embed_dim=64, patch_size=8):
        super().__init__()
        assert patch_size == 8
        self.patch_size, self.dim = patch_size, embed_dim
        self.proj = nn.Sequential(nn.Conv2d(in_chans, embed_dim // 4, 6, 2, 2), nn.ReLU(), nn.Conv2d(embed_dim // 4, embed_dim // 2, 6, 2, 2),
                                  nn.ReLU(), nn.Conv2d(embed_dim // 2, embed_dim, 6, 2, 2))
        self.ffn_with_coord = nn.Sequential(nn.Conv2d(2 * embed_dim, 2 * embed_dim, 1), nn.ReLU(), nn.Conv2d(2 * embed_dim, 2 * embed_dim, 1))
        self.norm = nn.LayerNorm(2 * embed_dim)

    def forward(self, x):
        S, _, H, W = x.shape
        p = self.patch_size
        x = F.pad(x, (0, (-W) % p, 0, (-H) % p))
        x = self.proj(x)
        h3, w3 = x.shape[-2:]
        centres = coords_grid(1, h3, w3, x.device) * p + p / 2
        code = sine_embedding(centres.flatten(2).transpose(1, 2), self.dim).transpose(1, 2).reshape(1, self.dim, h3, w3)
        x = self.ffn_with_coord(torch.cat([x, code.to(x.dtype).expand(S, -1, -1, -1)], dim=1))
        return self.norm(x.flatten(2).transpose(1, 2)), (h3, w3)


class TokenCrossAttention(nn.Module):
    """Pre-norm cross-attention + FFN of the latent tokens against the patch tokens of one cost map (the perceiver's input layer)."""

    def __init__(self, q_dim, kv_dim, heads=8):
        super().__init__()
        self.heads = heads
        self.norm1, self.norm2 = nn.LayerNorm(q_dim), nn.LayerNorm(q_dim)
        self.q, self.k, self.v, self.proj = nn.Linear(q_dim, q_dim), nn.Linear(kv_dim, q_dim), nn.Linear(kv_dim, q_dim), nn.Linear(q_dim, q_dim)
        self.ffn = nn.Sequential(nn.Linear(q_dim, q_dim), nn.GELU(), nn.Linear(q_dim, q_dim))

    def forward(self, query, tokens):
        query = query.expand(tokens.shape[0], -1, -1)
        x = query + self.proj(_mha(self.q(self.norm1(query)), self.k(tokens), self.v(tokens), self.heads))
        return x + self.ffn(self.norm2(x))


class LatentSelfAttention(nn.Module):
    """Self-attention among the K latent tokens of one source pixel."""

    def __init__(self, dim, heads=8):
        super().__init__()
        self.heads = heads
        self.norm1, self.norm2 = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.q, self.k, self.v, self.proj = (nn.Linear(dim, dim) for _ in range(4))
        self.ffn = nn.Sequential(nn.Linear(dim, dim), nn.GELU(), nn.Linear(dim, dim))

    def forward(self, x):
        y = self.norm1(x)
        x = x + self.proj(_mha(self.q(y), self.k(y), self.v(y), self.heads))
        return x + self.ffn(self.norm2(x))


class _ContextAttn(nn.Module):
    """Attention across the H1 x W1 source pixels for one latent index (FlowFormer's "vertical" layers): queries / keys see the token, a
    ``vert_c_dim``-channel projection of the context feature and the pixel's position code; values see the token only.  ``ws`` > 1:
    windows (Twins LSA form); ``ws`` = 0: global with sr x sr sub-sampled keys / values (GSA form)."""

    def __init__(self, dim, heads, ws, sr, vert_c_dim):
        super().__init__()
        self.heads, self.ws, self.cq = heads, ws, dim + vert_c_dim
        self.context_proj = nn.Linear(256, vert_c_dim)
        self.q, self.k, self.v, self.proj = nn.Linear(self.cq, dim), nn.Linear(self.cq, dim), nn.Linear(dim, dim), nn.Linear(dim, dim)
        if not ws:
            self.sr_key, self.sr_value = nn.Conv2d(self.cq, self.cq, sr, sr), nn.Conv2d(dim, dim, sr, sr)
            self.norm_key, self.norm_value = nn.LayerNorm(self.cq), nn.LayerNorm(dim)

    def forward(self, x, size, context):
        B, N, C = x.shape
        H, W = size
        ctx = self.context_proj(context.flatten(2).transpose(1, 2).to(x.dtype))                  # [b, N, vert_c_dim]
        ctx = ctx.repeat_interleave(B // ctx.shape[0], dim=0) if ctx.shape[0] != B else ctx
        code = sine_embedding(coords_grid(1, H, W, x.device).flatten(2).transpose(1, 2), self.cq).to(x.dtype)
        xq = torch.cat([x, ctx], dim=-1) + code
        if self.ws:
            qw, (Hp, Wp) = _to_windows(xq.view(B, H, W, self.cq), self.ws)
            vw, _ = _to_windows(x.view(B, H, W, C), self.ws)
            out = _from_windows(_mha(self.q(qw), self.k(qw), self.v(vw), self.heads), B, H, W, Hp, Wp, self.ws)
        else:
            ks = self.norm_key(self.sr_key(xq.transpose(1, 2).reshape(B, self.cq, H, W)).flatten(2).transpose(1, 2))
            vs = self.norm_value(self.sr_value(x.transpose(1, 2).reshape(B, C, H, W)).flatten(2).transpose(1, 2))
            out = _mha(self.q(xq), self.k(ks), self.v(vs), self.heads)
        return self.proj(out)


class VerticalLayer(nn.Module):
    def __init__(self, dim, vert_c_dim):
        super().__init__()
        self.local_block = TwinsBlock(dim, _ContextAttn(dim, 8, 7, 4, vert_c_dim))
        self.global_block = TwinsBlock(dim, _ContextAttn(dim, 8, 0, 4, vert_c_dim))

    def forward(self, x, size, context):
        return self.global_block(self.local_block(x, size, context), size, context)


class CostPerceiverEncoder(nn.Module):
    """cost volume ``[B, heads, H1, W1, H2, W2]`` -> latent cost memory ``[B * H1 * W1, K, D]`` (+ the cost maps the decoder looks up)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.patch_embed = PatchEmbed(cfg.cost_heads_num, cfg.cost_latent_input_dim, cfg.patch_size)
        self.latent_tokens = nn.Parameter(torch.randn(1, cfg.cost_latent_token_num, cfg.cost_latent_dim))
        self.input_layer = TokenCrossAttention(cfg.cost_latent_dim, 2 * cfg.cost_latent_input_dim)
        self.encoder_layers = nn.ModuleList(LatentSelfAttention(cfg.cost_latent_dim) for _ in range(cfg.encoder_depth))
        self.vertical_encoder_layers = nn.ModuleList(VerticalLayer(cfg.cost_latent_dim, cfg.vert_c_dim) for _ in range(cfg.encoder_depth))

    def forward(self, cost_volume, context):
        B, heads, H1, W1, H2, W2 = cost_volume.shape
        K = self.cfg.cost_latent_token_num
        cost_maps = cost_volume.permute(0, 2, 3, 1, 4, 5).reshape(B * H1 * W1, heads, H2, W2)
        tokens, _ = self.patch_embed(cost_maps)
        x = self.input_layer(self.latent_tokens.to(tokens.dtype), tokens)
        short_cut = x
        for layer, vertical in zip(self.encoder_layers, self.vertical_encoder_layers):
            x = layer(x)
            x = x.view(B, H1 * W1, K, -1).transpose(1, 2).reshape(B * K, H1 * W1, -1)
            x = vertical(x, (H1, W1), context)
            x = x.view(B, K, H1 * W1, -1).transpose(1, 2).reshape(B * H1 * W1, K, -1)
        return (x + short_cut if self.cfg.cost_encoder_res else x), cost_maps


class MemoryEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.feat_encoder = TwinsSVTLargeStem()
        self.channel_convertor = nn.Conv2d(cfg.encoder_latent_dim, cfg.encoder_latent_dim, 1, bias=False)
        self.cost_perceiver_encoder = CostPerceiverEncoder(cfg)

    def corr(self, fmap1, fmap2):
        """The all-pairs volume as FlowFormer's ``MemoryEncoder.corr`` defines it (reached from flownet.py:26; restated in oracle/corr.py)."""
        B, D, H, W = fmap1.shape
        h = self.cfg.cost_heads_num
        a = fmap1.view(B, h, D // h, H * W).transpose(2, 3)
        b = fmap2.view(B, h, D // h, H * W).transpose(2, 3)
        return torch.einsum("bhid,bhjd->bhij", a, b).view(B, h, H, W, H, W)

    def forward(self, img1, img2, context):
        feats = self.channel_convertor(self.feat_encoder(torch.cat([img1, img2], dim=0)))
        B = feats.shape[0] // 2
        return self.cost_perceiver_encoder(self.corr(feats[:B], feats[B:]), context)


# ----------------------------------------------------------------------------------------------------------- decoder
class SepConvGRU(nn.Module):
    """RAFT's separable ConvGRU (1x5 then 5x1), the update cell of both branches (covhead.py:29-31)."""

    def __init__(self, hidden_dim=128, input_dim=384):
        super().__init__()
        c = hidden_dim + input_dim
        self.convz1, self.convr1, self.convq1 = (nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2)) for _ in range(3))
        self.convz2, self.convr2, self.convq2 = (nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0)) for _ in range(3))

    def forward(self, h, x):
        for cz, cr, cq in ((self.convz1, self.convr1, self.convq1), (self.convz2, self.convr2, self.convq2)):
            hx = torch.cat([h, x], dim=1)
            z, r = torch.sigmoid(cz(hx)), torch.sigmoid(cr(hx))
            q = torch.tanh(cq(torch.cat([r * h, x], dim=1)))
            h = (1 - z) * h + z * q
        return h


class MotionEncoder(nn.Module):
    def __init__(self, cor_planes):
        super().__init__()
        self.convc1, self.convc2 = nn.Conv2d(cor_planes, 256, 1), nn.Conv2d(256, 192, 3, padding=1)
        self.convf1, self.convf2 = nn.Conv2d(2, 128, 7, padding=3), nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 192, 128 - 2, 3, padding=1)

    def forward(self, flow, corr):
        cor = F.relu(self.convc2(F.relu(self.convc1(corr))))
        flo = F.relu(self.convf2(F.relu(self.convf1(flow))))
        return torch.cat([F.relu(self.conv(torch.cat([cor, flo], dim=1))), flow], dim=1)


class GMAAttention(nn.Module):
    """GMA: one-head self-similarity of the context features, ``[B, 1, N, N]`` (computed once per pair, covhead.py:80)."""

    def __init__(self, dim=128, dim_head=128):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.to_qk = nn.Conv2d(dim, 2 * dim_head, 1, bias=False)

    def forward(self, fmap):
        q, k = self.to_qk(fmap).flatten(2).chunk(2, dim=1)                        # [B, d, N] each
        return torch.softmax(self.scale * q.transpose(1, 2) @ k, dim=-1)[:, None]


class GMAAggregate(nn.Module):
    def __init__(self, dim=128):
        super().__init__()
        self.to_v = nn.Conv2d(dim, dim, 1, bias=False)
        self.gamma = nn.Parameter(torch.full((1,), 0.1))

    def forward(self, attn, fmap):
        B, C, H, W = fmap.shape
        out = attn[:, 0] @ self.to_v(fmap).flatten(2).transpose(1, 2)              # [B, N, C]
        return fmap + self.gamma * out.transpose(1, 2).reshape(B, C, H, W)


class GMAUpdateBlock(nn.Module):
    def __init__(self, cfg, hidden_dim=128):
        super().__init__()
        self.encoder = MotionEncoder(cfg.query_latent_dim + 81 * cfg.cost_heads_num)
        self.gru = SepConvGRU(hidden_dim, 128 + hidden_dim + hidden_dim)
        self.flow_head = nn.Sequential(nn.Conv2d(hidden_dim, 256, 3, padding=1), nn.ReLU(), nn.Conv2d(256, 2, 3, padding=1))
        self.mask = nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(), nn.Conv2d(256, 64 * 9, 1))
        self.aggregator = GMAAggregate(128)


class CovHead(nn.Module):
    """covhead.py:8-21."""

    def __init__(self, input_dim=128, hidden_dim=256):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, hidden_dim // 2, 3, padding=1)
        self.conv3 = nn.Conv2d(hidden_dim // 2, hidden_dim // 4, 3, padding=1)
        self.conv4 = nn.Conv2d(hidden_dim // 4, 2, 3, padding=1)

    def forward(self, x):
        x = self.conv2(F.relu(self.conv1(x)))
        return self.conv4(F.relu(self.conv3(x)))


class CovUpdateBlock(nn.Module):
    """covhead.py:24-43."""

    def __init__(self, hidden_dim=128):
        super().__init__()
        self.gru = SepConvGRU(hidden_dim, 128 + hidden_dim + hidden_dim)
        self.cov_head = CovHead(hidden_dim, 256)
        self.mask = nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(), nn.Conv2d(256, 64 * 9, 1))

    def forward(self, covs_net, inp_cat):
        covs_net = self.gru(covs_net, inp_cat)
        return covs_net, self.cov_head(covs_net), 0.25 * self.mask(covs_net)


class DecoderCrossAttention(nn.Module):
    """One query token per source pixel (its 9x9 window, encoded, plus the code of where it currently points) attends to the pixel's K
    latent cost tokens; keys / values are projected once and handed back for the next iteration (covhead.py:100-102)."""

    def __init__(self, cfg, heads=8):
        super().__init__()
        d, m = cfg.query_latent_dim, cfg.cost_latent_dim
        self.heads, self.dim, self.flow_token = heads, d, cfg.add_flow_token
        self.norm1, self.norm2 = nn.LayerNorm(d), nn.LayerNorm(d)
        self.q, self.k, self.v, self.proj = nn.Linear(d, d), nn.Linear(m, d), nn.Linear(m, d), nn.Linear(2 * d, d)
        self.ffn = nn.Sequential(nn.Linear(d, d), nn.GELU(), nn.Linear(d, d))

    def forward(self, query, key, value, memory, coords1):
        B, _, H, W = coords1.shape
        short_cut = query
        q_in = self.norm1(query)
        if self.flow_token:
            q_in = q_in + sine_embedding(coords1.permute(0, 2, 3, 1).reshape(B * H * W, 1, 2), self.dim).to(query.dtype)
        if key is None:
            key, value = self.k(memory), self.v(memory)
        x = short_cut + self.proj(torch.cat([_mha(self.q(q_in), key, value, self.heads), short_cut], dim=2))
        return x + self.ffn(self.norm2(x)), key, value


class MemoryDecoderLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cross_attend = DecoderCrossAttention(cfg)

    def forward(self, query, key, value, memory, coords1, size, dim):
        x, key, value = self.cross_attend(query, key, value, memory, coords1)
        B, _, H, W = size
        return x.view(B, H, W, dim).permute(0, 3, 1, 2), key, value


class MemoryCovDecoder(nn.Module):
    """``MemoryCovDecoder`` (covhead.py:46-140) on top of the public ``MemoryDecoder``'s members."""

    def __init__(self, cfg, decoder_dtype=torch.float32):
        super().__init__()
        self.cfg, self.dim, self.depth, self.decoder_dtype = cfg, cfg.query_latent_dim, cfg.decoder_depth, decoder_dtype
        d = self.dim
        self.flow_token_encoder = nn.Sequential(nn.Conv2d(81 * cfg.cost_heads_num, d, 1), nn.GELU(), nn.Conv2d(d, d, 1))
        self.proj = nn.Conv2d(256, 256, 1)
        self.decoder_layer = MemoryDecoderLayer(cfg)
        self.update_block = GMAUpdateBlock(cfg, 128)
        self.att = GMAAttention(128, 128)
        self.cov_update = CovUpdateBlock(128)
        r = 4
        dy, dx = torch.meshgrid(torch.linspace(-r, r, 2 * r + 1), torch.linspace(-r, r, 2 * r + 1), indexing="ij")
        self.register_buffer("delta", torch.stack([dy, dx], dim=-1).view(1, 2 * r + 1, 2 * r + 1, 2), persistent=False)
        for m in (self.att, self.decoder_layer, self.flow_token_encoder, self.update_block, self.cov_update):
            m.to(dtype=decoder_dtype)

    # -- the two methods the HIP library replaces, as published --------------------------------------------------------------
    def encode_flow_token(self, cost_maps, coords):
        """RAFT-style 9x9 bilinear window around ``coords`` in every source pixel's own cost map; note the published quirk: the window
        offsets are stacked (dy, dx) and added to (x, y), so channel k = 9 i + j samples (x + i - 4, y + j - 4) (oracle/corr.py)."""
        B, _, H, W = coords.shape
        centre = coords.permute(0, 2, 3, 1).reshape(B * H * W, 1, 1, 2)
        pts = centre + self.delta.to(coords.dtype)
        h2, w2 = cost_maps.shape[-2:]
        grid = torch.stack([2 * pts[..., 0] / (w2 - 1) - 1, 2 * pts[..., 1] / (h2 - 1) - 1], dim=-1)
        win = F.grid_sample(cost_maps, grid, align_corners=True)
        return win.view(B, H, W, -1).permute(0, 3, 1, 2)

    def upsample_flow(self, flow, mask):
        """Convex 8x upsampling (RAFT): softmax over 9 neighbours of 8 * flow."""
        N, C, H, W = flow.shape
        mask = torch.softmax(mask.view(N, 1, 9, 8, 8, H, W), dim=2)
        nb = F.unfold(8 * flow, [3, 3], padding=1).view(N, C, 9, 1, 1, H, W)
        return (mask * nb).sum(dim=2).permute(0, 1, 4, 2, 5, 3).reshape(N, C, 8 * H, 8 * W)

    # -- covhead.py:60-140 ---------------------------------------------------------------------------------------------------
    def forward(self, cost_memory, context, cost_maps):
        dt = self.decoder_dtype
        cost_memory = cost_memory.to(dt)
        B, _, H, W = context.shape
        flow_c0 = coords_grid(B, H, W, context.device)
        flow_c1, cov_c0, cov_c1 = flow_c0.clone(), flow_c0, flow_c0.clone()
        ctx = self.proj(context)
        net, inp = torch.split(ctx, [128, 128], dim=1)
        flow_net = net.tanh().to(dt)
        cov_net = flow_net.clone()
        inp = inp.relu().to(dt)
        attention = self.att(inp)
        size, key, value = flow_net.shape, None, None
        flow_up = cov_up = None
        for _ in range(self.depth):
            flow_c1 = flow_c1.detach()
            flow = (flow_c1 - flow_c0).to(dt)
            cost_forward = self.encode_flow_token(cost_maps, flow_c1).to(dt)                            # fp32 island (:90-93)
            query = self.flow_token_encoder(cost_forward).permute(0, 2, 3, 1).reshape(B * H * W, 1, self.dim)
            cost_global, key, value = self.decoder_layer(query, key, value, cost_memory, flow_c1.to(dt), size, self.dim)
            corr = torch.cat([cost_global, cost_forward], dim=1)
            motion = self.update_block.encoder(flow, corr)
            inp_cat = torch.cat([inp, motion, self.update_block.aggregator(attention, motion)], dim=1)
            flow_net = self.update_block.gru(flow_net, inp_cat)
            delta_flow, up_mask = self.update_block.flow_head(flow_net), self.update_block.mask(flow_net)
            cov_net, delta_cov, cov_mask = self.cov_update(cov_net, inp_cat)
            flow_c1 = flow_c1 + delta_flow.float()                                                      # fp32 island (:119-126)
            flow_up = self.upsample_flow(flow_c1 - flow_c0, 0.25 * up_mask.float())
            cov_c1 = cov_c1 + delta_cov.float()                                                         # fp32 island (:128-135)
            cov_up = self.upsample_flow(cov_c1 - cov_c0, cov_mask.float())
        return (flow_up, flow_c1 - flow_c0), (cov_up, cov_c1 - cov_c0)


# ----------------------------------------------------------------------------------------------------------- the network
class FlowFormerCovHost(nn.Module):
    """``FlowFormerCov`` (flownet.py:9-44): context encoder + memory encoder in ``encoder_dtype``, decoder in ``decoder_dtype``."""

    def __init__(self, cfg=None, encoder_dtype=torch.float32, decoder_dtype=torch.float32):
        super().__init__()
        self.cfg = cfg or demo_cfg()
        self.enc_dtype = encoder_dtype
        self.context_encoder = TwinsSVTLargeStem().to(dtype=encoder_dtype)
        self.memory_encoder = MemoryEncoder(self.cfg).to(dtype=encoder_dtype)
        self.memory_decoder = MemoryCovDecoder(self.cfg, decoder_dtype)
        self.tame()

    @torch.no_grad()
    def tame(self, step: float = 0.05):
        """Random weights, not trained ones: shrink the two heads that feed the recurrence so that twelve iterations move the matches by
        pixels, not by image widths (keeps the lookups inside the maps and the activations finite)."""
        for head in (self.memory_decoder.update_block.flow_head[-1], self.memory_decoder.cov_update.cov_head.conv4):
            head.weight.mul_(step)
            head.bias.zero_()

    def forward(self, image1, image2):
        image1 = (2 * image1 - 1.0).to(self.enc_dtype)
        image2 = (2 * image2 - 1.0).to(self.enc_dtype)
        context = self.context_encoder(image1)
        cost_memory, cost_maps = self.memory_encoder(image1, image2, context)
        return self.memory_decoder(cost_memory, context.float(), cost_maps.float())

    @torch.no_grad()
    def inference(self, image1, image2):
        padder = InputPadder(image1.shape)
        image1, image2 = padder.pad(image1, image2)
        flow_pre, cov_pre = self.forward(image1, image2)
        return padder.unpad(flow_pre[0]), torch.exp(2 * padder.unpad(cov_pre[0]))


def parameter_count(m: nn.Module) -> int:
    return sum(p.numel() for p in m.parameters())
