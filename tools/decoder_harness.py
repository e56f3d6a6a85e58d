"""Decoder-loop harness: the HIP window lookup and convex upsampling interleaved with real PyTorch-ROCm kernels.

Follows the loop of ``MemoryCovDecoder.forward`` (``Module/Network/FlowFormerCov/covhead.py:60-140``) statement by
statement — per iteration: lookup (``encode_flow_token``, :92) -> token encoder (:96) -> cross-attention with the latent
cost memory (:100-103) -> motion encoder / aggregator (:106-107) -> flow GRU + heads (:112-114) -> covariance GRU + heads
(:117) -> fp32 convex upsampling of flow (:119-126) and log-sigma (:128-135) — with

* the HIP kernels where the reference says "MUST run in fp32": ``mv_corr_lookup`` for the lookup, ``mv_convex_upsample`` for
  both upsamplings (the second one with the fused ``exp(2*cov)`` of ``flownet.py:44`` on the last iteration);
* ``CovHead`` / ``CovUpdateBlock`` restated from the in-tree ``covhead.py:8-43`` (they cannot be imported: the file's first
  lines import the absent FlowFormer submodule), ``SepConvGRU`` restated from the public RAFT update block they use;
* randomly initialised STAND-INS of matching shape for the blocks that only exist in the absent submodule
  (``flow_token_encoder``, ``decoder_layer``, ``update_block.encoder / aggregator / flow_head / mask``): their arithmetic is
  not FlowFormer's, their kernel mix (1x1 / 3x3 / 7x7 / separable convolutions, one batched attention, concatenations, dtype
  casts) is.

Purpose: ``bench.py`` pre-bakes the twelve coordinate sets, so its lookups run back to back; here every
lookup waits for the previous iteration's GRU, as in the real network — the interleaved figure is reported next to the
back-to-back one.  The harness is measurement plumbing, not part of the drop-in (the real decoder is hooked through
``plugins.install_flowformer_hooks``).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from macvo_amd import ops


def _range(name: str):
    """Profiler range with the reference's names (``covhead.py:90-135`` wraps the same statements in ``torch.cuda.nvtx.range``;
    on ROCm these are roctx ranges, visible to ``rocprofv3 --marker-trace``).  A no-op context when the binding is unavailable."""
    try:
        return torch.cuda.nvtx.range(name)
    except Exception:  # noqa: BLE001
        import contextlib

        return contextlib.nullcontext()


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
        self.gru = SepConvGRU(hidden_dim=hidden_dim, input_dim=128 + hidden_dim + hidden_dim)
        self.cov_head = CovHead(hidden_dim, hidden_dim=256)
        self.mask = nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(256, 64 * 9, 1, padding=0))

    def forward(self, covs_net, inp_cat):
        covs_net = self.gru(covs_net, inp_cat)
        return covs_net, self.cov_head(covs_net), 0.25 * self.mask(covs_net)


class DecoderLoopHarness(nn.Module):
    def __init__(self, dim: int = 64, latent_tokens: int = 8, latent_dim: int = 128, radius: int = 4,
                 dec_dtype: torch.dtype = torch.bfloat16, depth: int = 12):
        super().__init__()
        self.radius, self.depth, self.dtype, self.dim = radius, depth, dec_dtype, dim
        kk = (2 * radius + 1) ** 2
        self.proj = nn.Conv2d(256, 256, 1)                                                       # decoder.proj
        # ---- stand-ins (absent submodule) ----
        self.flow_token_encoder = nn.Sequential(nn.Conv2d(kk, dim, 1), nn.GELU(), nn.Conv2d(dim, dim, 1))
        self.q_proj, self.k_proj, self.v_proj = nn.Linear(dim, dim), nn.Linear(latent_dim, dim), nn.Linear(latent_dim, dim)
        self.attn_out = nn.Linear(dim, 128)
        self.enc_c1, self.enc_c2 = nn.Conv2d(128 + kk, 256, 1), nn.Conv2d(256, 192, 3, padding=1)      # motion encoder
        self.enc_f1, self.enc_f2 = nn.Conv2d(2, 128, 7, padding=3), nn.Conv2d(128, 64, 3, padding=1)
        self.enc_out = nn.Conv2d(192 + 64, 126, 3, padding=1)
        self.aggregate = nn.Conv2d(128, 128, 1)                                                   # GMA aggregator stand-in
        self.flow_gru = SepConvGRU(128, 128 + 128 + 128)
        self.flow_head = nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(256, 2, 3, padding=1))
        self.flow_mask = nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(256, 64 * 9, 1))
        # ---- in-tree (restated) ----
        self.cov_update = CovUpdateBlock(128)
        for m in (self.flow_token_encoder, self.q_proj, self.k_proj, self.v_proj, self.attn_out, self.enc_c1, self.enc_c2,
                  self.enc_f1, self.enc_f2, self.enc_out, self.aggregate, self.flow_gru, self.flow_head, self.flow_mask,
                  self.cov_update):
            m.to(dtype=dec_dtype)
        self.hip_events: list = []
        self.trace: list | None = None        # set to [] to record (coords1, lookup tokens, flow8, up_mask, flow_up) per iteration
        self.last: dict = {}

    @torch.no_grad()
    def forward(self, cost_maps: torch.Tensor, cost_memory: torch.Tensor, context: torch.Tensor, time_hip: bool = False):
        """cost_maps ``[B*N, 1, H8, W8]`` fp32 (the HIP cost volume), cost_memory ``[B*N, tokens, latent_dim]``, context
        ``[B, 256, H8, W8]`` fp32 -> ``(flow_up [B,2,H,W], flow8), (cov_up = exp(2*log-sigma) [B,2,H,W], cov8)``."""
        dt = self.dtype
        B, _, h8, w8 = context.shape
        N = h8 * w8
        ev = self.hip_events = []

        def hip(fn, *a, **k):
            if not time_hip:
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            ev.append((fn.__name__, e0, e1))
            return out

        cost_memory = cost_memory.to(dt)
        ys, xs = torch.meshgrid(torch.arange(h8, device=context.device), torch.arange(w8, device=context.device), indexing="ij")
        coords0 = torch.stack([xs, ys]).float()[None].repeat(B, 1, 1, 1)                        # initialize_flow
        flow_c1, cov_c0, cov_c1 = coords0.clone(), coords0, coords0.clone()
        ctx = self.proj(context)
        flow_net, flow_inp = torch.split(ctx, [128, 128], dim=1)
        flow_net = flow_net.tanh().to(dt)
        fcov_net = flow_net.clone()
        flow_inp = flow_inp.relu().to(dt)
        k_mem, v_mem = self.k_proj(cost_memory), self.v_proj(cost_memory)                       # decoder_layer keeps K / V (:100)
        flow_up = cov_up = None
        for it in range(self.depth):
            flow = (flow_c1 - coords0).to(dt)
            with _range("Encode Flow Token"):
                tokens = hip(ops.corr_lookup, cost_maps, flow_c1, self.radius)                   # :92  MUST run in fp32
                if self.trace is not None:
                    self.trace.append(dict(coords=flow_c1.clone(), tokens=tokens.clone()))
                cost_forward = tokens.to(dt)
            with _range("CNN Encoder"):
                query = self.flow_token_encoder(cost_forward)                                    # :96
                query = query.permute(0, 2, 3, 1).reshape(B * N, 1, self.dim)
            with _range("Cross Attention"):
                att = torch.softmax(self.q_proj(query) @ k_mem.transpose(1, 2) / self.dim ** 0.5, dim=-1)   # :100-102
                cost_global = self.attn_out(att @ v_mem).view(B, h8, w8, 128).permute(0, 3, 1, 2)
                corr = torch.cat([cost_global, cost_forward], dim=1)                              # :103
            with _range("GMA Update Block"):
                cor = F.relu(self.enc_c2(F.relu(self.enc_c1(corr))))                              # :106 motion encoder
                flo = F.relu(self.enc_f2(F.relu(self.enc_f1(flow))))
                motion_feat = torch.cat([F.relu(self.enc_out(torch.cat([cor, flo], dim=1))), flow], dim=1)
                motion_global = self.aggregate(motion_feat)                                       # :107
            inp_cat = torch.cat([flow_inp, motion_feat, motion_global], dim=1)                    # :109
            with _range("Flow Update Block"):
                flow_net = self.flow_gru(flow_net, inp_cat)                                       # :112-114
                delta_flow, up_mask = self.flow_head(flow_net), self.flow_mask(flow_net)
            with _range("Cov Update Block"):
                fcov_net, delta_cov, cov_mask = self.cov_update(fcov_net, inp_cat)                # :117
            with _range("Flow Upsample"):
                flow_c1 = flow_c1 + delta_flow.float()                                            # :121-126, fp32
                um = up_mask.contiguous()            # the mask in the decoder's own dtype: mv_convex_upsample_m widens it in registers (no 11 -> 22 MB copy)
                flow_up = hip(ops.convex_upsample, flow_c1 - coords0, um, mask_scale=0.25)
                if self.trace is not None:
                    self.trace[-1].update(flow8=(flow_c1 - coords0).clone(), up_mask=um.clone(), flow_up=flow_up.clone())
            with _range("Cov Upsample"):
                cov_c1 = cov_c1 + delta_cov.float()                                               # :130-135, fp32
                last = it == self.depth - 1
                cm = cov_mask.contiguous()
                cov_up = hip(ops.convex_upsample, cov_c1 - cov_c0, cm, mask_scale=1.0, exp2_out=last)
        # what the hot path takes over (pipeline.FrameInputs, the 1/8-resolution alternative): last iteration's fields + masks
        self.last = dict(flow8=(flow_c1 - coords0).contiguous(), cov8=(cov_c1 - cov_c0).contiguous(), up_mask=um.float(), cov_mask=cm.float())   # FrameInputs' contract: fp32 masks
        return (flow_up, flow_c1 - coords0), (cov_up, cov_c1 - cov_c0)

    def hip_times_us(self) -> dict:
        """After a ``time_hip=True`` forward + synchronize: mean microseconds per HIP op kind."""
        acc: dict = {}
        for name, e0, e1 in self.hip_events:
            acc.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3)
        return {k: sum(v) / len(v) for k, v in acc.items()}
