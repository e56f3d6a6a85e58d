"""Torch-tensor front doors of the C ABI (``include/macvo_hip.h``).

PyTorch is only the allocator / stream provider here: each function validates shapes, allocates the
outputs, and hands raw device pointers plus the *current* HIP stream to ``libmacvo_hip.so``.  Nothing is
computed in torch and there is no fallback path — a missing library or a non-GPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from . import _lib as L

_DT = {torch.float32: L.MV_F32, torch.float16: L.MV_F16, torch.bfloat16: L.MV_BF16}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """Raw hipStream_t of torch's CURRENT stream on the current device.  `torch.cuda.current_stream()` builds a Python
    Stream object through several layers (~8 us per call, ~25 calls per frame); the raw getter is ~0.2 us."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise L.MacvoHipError(f"{name}: expected a GPU tensor (the HIP hot path has no CPU fallback)")
    if t.dtype != dtype:
        raise L.MacvoHipError(f"{name}: expected {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _u8(t: torch.Tensor | None) -> torch.Tensor | None:
    if t is None:
        return None
    if t.dtype == torch.bool:
        t = t.contiguous().view(torch.uint8)
    return _req(t, torch.uint8, "mask")


# ------------------------------------------------------------------------------------------- A5
def default_volume_precision() -> str:
    """THE default arithmetic of the fp32 cost volume — one value for ``corr_volume``, ``pipeline.HotPathConfig``,
    ``plugins.install_flowformer_hooks`` (hence every ``HIP_*`` frontend plugin, whose YAML key set is the reference's and cannot
    carry the choice) and ``bench.py``: ``"f16x2"``, the fastest form that keeps the fp32 parity bar
    (error <= ~2^-21 sum |a||b|, not narrower than the TF32 the reference runs this GEMM in, Frontend.py:275-277); shapes the
    streaming kernel does not cover fall back to "exact".  ``MACVO_HIP_VOLUME_PRECISION=exact|bf16x3|f16x2`` overrides it."""
    import os

    v = os.environ.get("MACVO_HIP_VOLUME_PRECISION", "f16x2")
    if v not in ("exact", "bf16x3", "f16x2"):
        raise L.MacvoHipError(f"MACVO_HIP_VOLUME_PRECISION={v!r}: expected exact | bf16x3 | f16x2")
    return v


def corr_volume(f1: torch.Tensor, f2: torch.Tensor, layout: str = "chw", out: torch.Tensor | None = None,
                precision: str | None = None) -> torch.Tensor:
    """All-pairs cost volume (FlowFormer ``MemoryEncoder.corr``; call site flownet.py:26-27).

    layout "chw": f1, f2 ``[B, C, H, W]`` (NCHW);  layout "hwc": ``[B, H, W, C]`` / ``[B, N, C]``.
    precision (fp32 inputs only; None = ``default_volume_precision()``, 16-bit inputs ignore it): "exact" = fp32 MFMA (bitwise fmaf chain); "f16x2" = rows scaled by a power of two into fp16's
    range, two fp16 pieces, three products (error <= ~2^-21 sum |a||b|: inside the parity bar, the fastest form); "bf16x3" = operands packed into three bf16 pieces
    (``volume_pack``) + the streaming six-product kernel on the 16-bit matrix pipe, fp32-class accuracy (same parity bar as
    "exact", not bitwise), either layout, shapes the kernel does not cover fall back to "exact"; "split3" / "split2" = the
    round-1/2 tile kernels over pre-split planes (6 / 3 products; "split2": relative error ~2^-16, finer than TF32, the class
    the reference's fast frontend allows itself) — both layout "hwc" only.
    Returns ``cost_maps [B*H1*W1, 1, H2, W2]`` float32 (layout "hwc" with 3-D inputs: ``[B*N1, 1, 1, N2]``).
    """
    lib = L.load()
    if f1.dtype not in _DT or f1.dtype != f2.dtype:
        raise L.MacvoHipError(f"corr_volume: unsupported dtypes {f1.dtype}/{f2.dtype}")
    f1 = _req(f1, f1.dtype, "f1")
    f2 = _req(f2, f2.dtype, "f2")
    if layout == "chw":
        B, Cc, H1, W1 = f1.shape
        _, _, H2, W2 = f2.shape
        lay = L.MV_LAYOUT_CHW
    elif layout == "hwc":
        if f1.dim() == 4:
            B, H1, W1, Cc = f1.shape
            _, H2, W2, _ = f2.shape
        else:
            B, N1_, Cc = f1.shape
            H1, W1, H2, W2 = 1, N1_, 1, f2.shape[1]
        lay = L.MV_LAYOUT_HWC
    else:
        raise ValueError(layout)
    N1, N2 = H1 * W1, H2 * W2
    if out is None:
        out = torch.empty((B * N1, 1, H2, W2), dtype=torch.float32, device=f1.device)
    dt = _DT[f1.dtype]
    p1, p2 = f1, f2
    if precision is None:
        precision = default_volume_precision() if f1.dtype == torch.float32 else "exact"
    if precision in _PACK_MODE:
        if f1.dtype != torch.float32:
            raise L.MacvoHipError(f"corr_volume: precision='{precision}' splits float32 inputs")
        if lib.mv_corr_volume_packed_supported(B, Cc, N1, N2, _PACK_MODE[precision]):
            pk1, pk2 = volume_pack(f1, f2, layout, mode=precision)
            return corr_volume_packed(pk1, pk2, B, Cc, N1, N2, out=out, mode=precision)
        precision = "exact"
    if precision in ("split3", "split2"):
        if f1.dtype != torch.float32 or lay != L.MV_LAYOUT_HWC:
            raise L.MacvoHipError(f"corr_volume: precision='{precision}' needs float32 inputs in layout 'hwc'")
        p1 = torch.empty((3,) + tuple(f1.shape), dtype=torch.bfloat16, device=f1.device)
        p2 = torch.empty((3,) + tuple(f2.shape), dtype=torch.bfloat16, device=f2.device)
        L.check(lib.mv_split_bf16x3(f1.data_ptr(), p1.data_ptr(), f1.numel(), _stream()), "mv_split_bf16x3")
        L.check(lib.mv_split_bf16x3(f2.data_ptr(), p2.data_ptr(), f2.numel(), _stream()), "mv_split_bf16x3")
        dt = L.MV_BF16X3 if precision == "split3" else L.MV_BF16X2
    elif precision != "exact":
        raise ValueError(precision)
    L.check(lib.mv_corr_volume(p1.data_ptr(), p2.data_ptr(), out.data_ptr(), B, Cc, N1, N2, dt, lay, _stream()),
            "mv_corr_volume")
    return out


_PACK_MODE = {"bf16x3": L.MV_PACK_BF16X3, "f16x2": L.MV_PACK_F16X2}


def volume_pack(f1: torch.Tensor, f2: torch.Tensor, layout: str = "chw", out: "tuple | None" = None, mode: str = "bf16x3",
                tiled_hw: "tuple | None" = None):
    """fp32 feature maps -> the packed operands of ``corr_volume_packed`` (``mv_volume_pack``: both maps in one launch,
    MFMA-fragment order, row N - 1 replicated past the edge).  mode "bf16x3": three bf16 pieces; "f16x2": two fp16 pieces of
    every value after a per-row power-of-two scaling (+ the table of row exponents).  Returns two uint8 tensors (opaque).
    ``tiled_hw=(H2, W2)``: operand 2 in 4 x 4-tile order (``mv_volume_pack_tiled``) -> the volume comes out tiled for
    ``corr_lookup(..., tiled=True)``."""
    md = _PACK_MODE[mode]
    lib = L.load()
    f1 = _req(f1, torch.float32, "f1")
    f2 = _req(f2, torch.float32, "f2")
    if layout == "chw":
        B, Cc = f1.shape[0], f1.shape[1]
        N1, N2 = f1[0, 0].numel(), f2[0, 0].numel()
    else:
        B, Cc = f1.shape[0], f1.shape[-1]
        N1, N2 = f1[0, ..., 0].numel(), f2[0, ..., 0].numel()
    n1, n2 = lib.mv_volume_pack_bytes(B, Cc, N1, md), lib.mv_volume_pack_bytes(B, Cc, N2, md)
    if n1 == 0 or n2 == 0:
        raise L.MacvoHipError("volume_pack: unsupported shape (C % 16 != 0?)")
    p1, p2 = out if out is not None else (torch.empty(n1, dtype=torch.uint8, device=f1.device), torch.empty(n2, dtype=torch.uint8, device=f1.device))
    assert p1.numel() >= n1 and p2.numel() >= n2
    lay = L.MV_LAYOUT_CHW if layout == "chw" else L.MV_LAYOUT_HWC
    if tiled_hw is not None:
        H2, W2 = tiled_hw
        assert H2 * W2 == N2
        L.check(lib.mv_volume_pack_tiled(f1.data_ptr(), f2.data_ptr(), p1.data_ptr(), p2.data_ptr(), B, Cc, N1, H2, W2, lay, md, _stream()),
                "mv_volume_pack_tiled")
    else:
        L.check(lib.mv_volume_pack(f1.data_ptr(), f2.data_ptr(), p1.data_ptr(), p2.data_ptr(), B, Cc, N1, N2, lay, md, _stream()),
                "mv_volume_pack")
    return p1, p2


def corr_volume_packed(pk1: torch.Tensor, pk2: torch.Tensor, B: int, C_: int, N1: int, N2: int, out: torch.Tensor | None = None,
                       mode: str = "bf16x3", free_cus: int = 0) -> torch.Tensor:
    """The cost volume ``[B*N1, 1, 1, N2]`` fp32 from two packed operands (``mv_corr_volume_packed``: six bf16 / three fp16 piece
    products, fp32 accumulate).  ``free_cus``: compute units left without a persistent workgroup (``mv_corr_volume_packed_shared``;
    same bits)."""
    lib = L.load()
    if out is None:
        out = torch.empty((B * N1, 1, 1, N2), dtype=torch.float32, device=pk1.device)
    L.check(lib.mv_corr_volume_packed_shared(pk1.data_ptr(), pk2.data_ptr(), out.data_ptr(), B, C_, N1, N2, _PACK_MODE[mode], int(free_cus), _stream()),
            "mv_corr_volume_packed_shared")
    return out


def last_volume_kernel() -> str:
    """Name of the kernel the last ``corr_volume`` call of this thread dispatched (diagnostics: dispatch tests, bench.py)."""
    return (L.load().mv_corr_volume_last_kernel() or b"").decode()


# ------------------------------------------------------------------------------------------- A23
def local_corr81(first: torch.Tensor, second: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """PWC-Net ``FunctionCorrelation(tenFirst, tenSecond)`` forward (pwc/correlation.py:277-325): ``[B,C,H,W]`` x2 ->
    ``[B,81,H,W]``, channel ``9*(dy+4)+(dx+4)`` = mean over C of ``first[y,x] * second[y+dy,x+dx]``, zero padded."""
    lib = L.load()
    first = _req(first, torch.float32, "first")
    second = _req(second, torch.float32, "second")
    if first.shape != second.shape or first.dim() != 4:
        raise L.MacvoHipError("local_corr81: first / second must be [B,C,H,W] tensors of the same shape")
    B, Cc, H, W = first.shape
    if out is None:
        out = torch.empty((B, 81, H, W), dtype=torch.float32, device=first.device)
    L.check(lib.mv_local_corr81(first.data_ptr(), second.data_ptr(), out.data_ptr(), B, Cc, H, W, _stream()),
            "mv_local_corr81")
    return out


# ------------------------------------------------------------------------------------------- A6
def corr_lookup(cost_maps: torch.Tensor, coords: torch.Tensor, radius: int = 4, out: torch.Tensor | None = None,
                tiled: bool = False, image_hw: "tuple | None" = None) -> torch.Tensor:
    """``encode_flow_token(cost_maps, coords)`` (covhead.py:92): ``[B*N1,1,H2,W2]``, ``[B,2,H1,W1]`` -> ``[B,(2r+1)^2,H1,W1]``.
    ``tiled``: the slices of ``cost_maps`` are stored in 4 x 4-cell tiles (``volume_pack(tiled_hw=...)`` + ``corr_volume_packed``, or
    ``corr_volume_out16(tiled=True)``); ``image_hw=(H2, W2)``: the image the slices cover when a tiled fp16 slice carries a padded
    last tile row (H2 % 4 != 0: ``cost_maps`` is then ``[B*N1, 1, 4 * ceil(H2 / 4), W2]``)."""
    lib = L.load()
    vol16 = cost_maps.dtype == torch.float16                    # Fast mode: the volume as `corr_volume_out16` stores it
    cost_maps = _req(cost_maps, torch.float16 if vol16 else torch.float32, "cost_maps")
    coords = _req(coords, torch.float32, "coords")
    B, two, H1, W1 = coords.shape
    assert two == 2
    BN, _, H2, W2 = cost_maps.shape
    if BN != B * H1 * W1:
        raise L.MacvoHipError("corr_lookup: cost_maps / coords shape mismatch")
    if image_hw is not None:
        if not (tiled and vol16) or image_hw[1] != W2 or lib.mv_tiled_slice_cells(int(image_hw[0]), W2) != H2 * W2:
            raise L.MacvoHipError("corr_lookup: image_hw belongs to a tiled fp16 volume of mv_tiled_slice_cells(H2, W2) cells per slice")
        H2 = int(image_hw[0])
    K = 2 * radius + 1
    if out is None:
        out = torch.empty((B, K * K, H1, W1), dtype=torch.float32, device=coords.device)
    name = "mv_corr_lookup" + ("_tiled" if tiled else "") + ("_vol16" if vol16 else "")
    L.check(getattr(lib, name)(cost_maps.data_ptr(), coords.data_ptr(), out.data_ptr(), B, H1, W1, H2, W2, radius, _stream()), name)
    return out


def fmap_tile_rows16(f: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """A 16-bit HWC feature map ``[B, H, W, C]`` with its pixel rows in 4 x 4-tile order (``mv_fmap_tile_rows16``): operand 2 of
    ``corr_volume_out16`` for a tiled Fast-mode volume."""
    lib = L.load()
    if f.dtype not in (torch.float16, torch.bfloat16) or f.dim() != 4:
        raise L.MacvoHipError("fmap_tile_rows16: a [B, H, W, C] float16 / bfloat16 feature map")
    f = _req(f, f.dtype, "f")
    B, H, W, Cc = f.shape
    cells = lib.mv_tiled_slice_cells(H, W)
    if cells == 0:
        raise L.MacvoHipError("fmap_tile_rows16: W % 4 != 0")
    if out is None:
        out = torch.empty((B, cells // W, W, Cc), dtype=f.dtype, device=f.device)      # (rows >= H of the last tile row: zero pixels)
    assert out.numel() >= B * cells * Cc and out.dtype == f.dtype
    L.check(lib.mv_fmap_tile_rows16(f.data_ptr(), out.data_ptr(), B, Cc, H, W, _stream()), "mv_fmap_tile_rows16")
    return out


def corr_volume_out16(f1: torch.Tensor, f2: torch.Tensor, out: torch.Tensor | None = None, tiled: bool = False,
                      scratch: torch.Tensor | None = None) -> torch.Tensor | None:
    """Fast mode (enc_dtype fp16 / bf16, MACVO_Fast.yaml:73-74): the volume in the encoder's 16-bit type — what ``einsum`` returns there and
    flownet.py:27 widens — rounded ONCE in the GEMM's epilogue.  ``f1, f2 [B, H, W, C]`` (HWC) fp16 / bf16 -> ``[B*H1*W1, 1, H2, W2]`` of that
    dtype; ``None`` for shapes outside the streaming kernel's domain (callers then cast ``corr_volume``'s fp32 result).
    ``tiled``: every slice in 4 x 4-cell tiles (``fmap_tile_rows16`` on ``f2`` first, into ``scratch`` if given) for
    ``corr_lookup(..., tiled=True)``."""
    lib = L.load()
    if f1.dtype not in (torch.float16, torch.bfloat16) or f1.dtype != f2.dtype or f1.dim() != 4:
        raise L.MacvoHipError("corr_volume_out16: [B, H, W, C] float16 / bfloat16 feature maps")
    f1, f2 = _req(f1, f1.dtype, "f1"), _req(f2, f2.dtype, "f2")
    B, H1, W1, Cc = f1.shape
    _, H2, W2, _ = f2.shape
    N1, N2 = H1 * W1, H2 * W2
    dt = _DT[f1.dtype]
    if tiled:                                                             # slices of mv_tiled_slice_cells(H2, W2) cells: a padded last tile row when H2 % 4 != 0
        N2 = lib.mv_tiled_slice_cells(H2, W2)
        if N2 == 0:
            return None
    if not lib.mv_corr_volume_out16_supported(B, Cc, N1, N2, dt, L.MV_LAYOUT_HWC):
        return None
    if out is None:
        out = torch.empty((B * N1, 1, N2 // W2, W2), dtype=f1.dtype, device=f1.device)
    if tiled:
        f2 = fmap_tile_rows16(f2, out=scratch)
    L.check(lib.mv_corr_volume_out16(f1.data_ptr(), f2.data_ptr(), out.data_ptr(), B, Cc, N1, N2, dt, L.MV_LAYOUT_HWC, _stream()),
            "mv_corr_volume_out16")
    return out


# ------------------------------------------------------------------------------------------- A8 / A2
@dataclass
class FrontendMaps:
    """The planes of IStereoDepth.Output / IMatcher.Output, each ``[1, C, H, W]`` float32 on the GPU."""
    depth: torch.Tensor
    depth_cov: torch.Tensor
    disparity: torch.Tensor
    disparity_cov: torch.Tensor
    bad_mask: torch.Tensor | None
    flow: torch.Tensor | None
    flow_cov: torch.Tensor | None


def frontend_epilogue(flow: torch.Tensor, cov: torch.Tensor, baseline: float, fx: float, cov_is_log: bool = True,
                      enforce_positive_disparity: bool = False, want_match: bool = True) -> FrontendMaps:
    """Network output ``flow, logcov [2,2,H,W]`` -> depth / match records (Frontend.py:183-200, flownet.py:44,
    StereoDepth.py:270-282, Matching.py:28-40) in one launch.  ``baseline``/``fx`` are Python floats exactly as
    ``frame.frame_baseline`` / ``frame.fx`` are in the reference."""
    lib = L.load()
    flow = _req(flow, torch.float32, "flow")
    cov = _req(cov, torch.float32, "cov")
    assert flow.shape == cov.shape and flow.shape[1] == 2
    assert flow.shape[0] == 2 or (flow.shape[0] == 1 and not want_match), "sample 0 = stereo pair, sample 1 = temporal pair"
    _, _, H, W = flow.shape
    dev = flow.device
    mk = lambda c: torch.empty((1, c, H, W), dtype=torch.float32, device=dev)  # noqa: E731
    disparity, disparity_cov, depth, depth_cov = mk(1), mk(1), mk(1), mk(1)
    bad = torch.empty((1, 1, H, W), dtype=torch.bool, device=dev) if enforce_positive_disparity else None
    mflow = mk(2) if want_match else None
    mcov = mk(3) if want_match else None
    bl_fx = float(baseline) * float(fx)            # python double product, rounded once to fp32 by ctypes
    L.check(lib.mv_frontend_epilogue(flow.data_ptr(), cov.data_ptr(), int(cov_is_log), H, W, bl_fx, bl_fx ** 2,
                                     disparity.data_ptr(), disparity_cov.data_ptr(), depth.data_ptr(),
                                     depth_cov.data_ptr(), _ptr(bad), _ptr(mflow), _ptr(mcov), _stream()),
            "mv_frontend_epilogue")
    return FrontendMaps(depth, depth_cov, disparity, disparity_cov, bad, mflow, mcov)


def convex_upsample(flow8: torch.Tensor, mask: torch.Tensor, mask_scale: float = 1.0, exp2_out: bool = False) -> torch.Tensor:
    """RAFT / FlowFormer ``upsample_flow`` (covhead.py:124-126,133-135): ``[B,2,h,w], [B,576,h,w] -> [B,2,8h,8w]``;
    ``exp2_out`` fuses the ``exp(2 * cov)`` of flownet.py:44.  ``mask`` may be fp32, fp16 or bf16 (the decoder's autocast type in Fast mode);
    the result is the fp32 formula on the widened values."""
    lib = L.load()
    flow8 = _req(flow8, torch.float32, "flow8")
    if mask.dtype not in _DT:
        raise TypeError(f"mask: fp32 / fp16 / bf16 expected, got {mask.dtype}")
    mask = _req(mask, mask.dtype, "mask")       # a 16-bit mask is read as it is (no widened copy) and widened in registers
    B, two, h, w = flow8.shape
    assert two == 2 and mask.shape == (B, 576, h, w)
    out = torch.empty((B, 2, 8 * h, 8 * w), dtype=torch.float32, device=flow8.device)
    L.check(lib.mv_convex_upsample_m(flow8.data_ptr(), mask.data_ptr(), _DT[mask.dtype], out.data_ptr(), B, h, w, float(mask_scale),
                                     int(exp2_out), _stream()), "mv_convex_upsample_m")
    return out


# ------------------------------------------------------------------------------------------- (f)2 cost patch-embed
class PatchEmbedWeights:
    """The three ``Conv2d`` layers of FlowFormer's cost ``PatchEmbed.proj`` (patch_size 8: 1 -> 16 -> 32 -> 64 channels, 6x6, stride 2,
    padding 2) packed once into the fragment order of ``mv_cost_patch_embed`` (16-bit weights, fp32 biases).

    ``operand``: the 16-bit type weights and activations are rounded to — ``"f16"`` (IEEE half: 11 significant bits, the mantissa of the
    TF32 the reference's fp32 configurations run their convolutions in, Frontend.py:275-277, and the type ``MACVO_Fast.yaml:73`` runs this
    encoder in) or ``"bf16"`` (for bf16 encoders).  None: by the dtype of ``w1`` — bf16 weights -> "bf16", fp16 / fp32 weights -> "f16"."""

    def __init__(self, w1, b1, w2, b2, w3, b3, operand: str | None = None):
        lib = L.load()
        if operand is None:
            operand = "bf16" if w1.dtype == torch.bfloat16 else "f16"
        if operand not in ("f16", "bf16"):
            raise L.MacvoHipError(f"PatchEmbedWeights: operand must be 'f16' or 'bf16', got {operand!r}")
        self.operand = operand
        self.operand_type = L.MV_F16 if operand == "f16" else L.MV_BF16
        ts = []
        for t, shp in zip((w1, b1, w2, b2, w3, b3), ((16, 1, 6, 6), (16,), (32, 16, 6, 6), (32,), (64, 32, 6, 6), (64,))):
            if tuple(t.shape) != shp:
                raise L.MacvoHipError(f"PatchEmbedWeights: expected a tensor of shape {shp}, got {tuple(t.shape)}")
            ts.append(_req(t.detach().float(), torch.float32, "patch-embed weight"))
        self.packed = torch.empty(int(lib.mv_patch_embed_packed_bytes()), dtype=torch.uint8, device=ts[0].device)
        L.check(lib.mv_patch_embed_pack(*[t.data_ptr() for t in ts], self.packed.data_ptr(), self.operand_type, _stream()), "mv_patch_embed_pack")
        self._keep = ts

    @classmethod
    def from_proj(cls, proj, operand: str | None = None) -> "PatchEmbedWeights":
        """``proj`` = the ``nn.Sequential(Conv2d, ReLU, Conv2d, ReLU, Conv2d)`` of a FlowFormer ``PatchEmbed`` (patch_size 8)."""
        convs = [m for m in proj if isinstance(m, torch.nn.Conv2d)]
        if len(convs) != 3 or any(c.kernel_size != (6, 6) or c.stride != (2, 2) or c.padding != (2, 2) or c.bias is None for c in convs):
            raise L.MacvoHipError("PatchEmbedWeights.from_proj: expected three Conv2d(k = 6, stride 2, padding 2, bias) layers (patch_size 8)")
        return cls(convs[0].weight, convs[0].bias, convs[1].weight, convs[1].bias, convs[2].weight, convs[2].bias, operand=operand)


def cost_patch_embed_supported(H2: int, W2: int) -> bool:
    return bool(L.load().mv_cost_patch_embed_supported(int(H2), int(W2)))


def cost_patch_embed(cost_maps: torch.Tensor, weights: PatchEmbedWeights, tokens: bool = False, out: torch.Tensor | None = None,
                     out_dtype: torch.dtype | None = None) -> torch.Tensor:
    """``PatchEmbed.proj(F.pad(cost_maps))`` for every slice in one launch: ``cost_maps [S, 1, H2, W2]`` (fp32 as ``corr_volume`` returns it, or the
    16-bit cells of ``corr_volume_out16`` when that type is ``weights.operand``) -> ``[S, 64, H2/8, W2/8]``, or with ``tokens`` the flattened-transposed
    form ``[S, H2/8 * W2/8, 64]``.  Result dtype: ``out_dtype`` / ``out.dtype`` if given, else the input's (what ``Conv2d`` returns) — fp32 or the operand
    type.  16-bit matrix pipe (``weights.operand``) with fp32 accumulation; the two intermediate maps never leave LDS, a 16-bit slice goes to the matrix
    pipe as it is and nothing is widened on either side.  Raises for slice sizes the kernel does not cover (``cost_patch_embed_supported``) and for
    dtype combinations it does not build: callers keep their PyTorch layers for those."""
    lib = L.load()
    op_dtype = torch.float16 if weights.operand == "f16" else torch.bfloat16
    if cost_maps.dtype not in (torch.float32, op_dtype):
        raise L.MacvoHipError(f"cost_patch_embed: cost_maps must be float32 or the operand type {op_dtype}, got {cost_maps.dtype}")
    cost_maps = _req(cost_maps, cost_maps.dtype, "cost_maps")
    S, H2, W2 = cost_maps.shape[0], cost_maps.shape[-2], cost_maps.shape[-1]
    if cost_maps.numel() != S * H2 * W2:
        raise L.MacvoHipError("cost_patch_embed: cost_maps must be [S, 1, H2, W2] (one head)")
    h, w = (H2 + 7) // 8, (W2 + 7) // 8
    shape = (S, h * w, 64) if tokens else (S, 64, h, w)
    if out is not None:
        out_dtype = out.dtype
    elif out_dtype is None:
        out_dtype = cost_maps.dtype
    if out_dtype not in (torch.float32, op_dtype) or (out_dtype != torch.float32 and cost_maps.dtype == torch.float32):
        raise L.MacvoHipError(f"cost_patch_embed: {cost_maps.dtype} slices -> {out_dtype} tokens is not built (operand {op_dtype})")
    if out is None:
        out = torch.empty(shape, dtype=out_dtype, device=cost_maps.device)
    elif tuple(out.shape) != shape or not out.is_contiguous():
        raise L.MacvoHipError(f"cost_patch_embed: out must be a contiguous tensor of shape {shape}")
    L.check(lib.mv_cost_patch_embed_t(cost_maps.data_ptr(), _DT[cost_maps.dtype], weights.packed.data_ptr(), out.data_ptr(), _DT[out_dtype], S, H2, W2,
                                      int(tokens), weights.operand_type, _stream()), "mv_cost_patch_embed_t")
    return out


# ------------------------------------------------------------------------------------------- A10 / A11
class KeypointCandidates:
    """Device-side result of the dense selector stage; ``finish`` applies the reference's CPU randperm."""

    def __init__(self, cand: torch.Tensor, count: torch.Tensor, stats: torch.Tensor, W: int):
        self.cand, self.count, self.stats, self.W = cand, count, stats, W
        self._n = None

    @property
    def n(self) -> int:
        if self._n is None:
            self._n = int(self.count[0].item())  # the one host sync of the selector (reference has two)
        return self._n

    def candidates_vu(self) -> torch.Tensor:
        """``torch.nonzero(point_mask)[:, 2:]`` — (v, u) int64, row-major order."""
        lin = self.cand[: self.n].long()
        return torch.stack([lin // self.W, lin % self.W], dim=1)

    def finish(self, numPoint: int, staging: torch.Tensor | None = None) -> torch.Tensor:
        """``selected[torch.randperm(n)[:numPoint]][..., 2:].roll(1, 1)`` (KeypointSelector.py:331-332,404-405).
        The permutation comes from the global CPU generator exactly as in the reference.  ``staging``: optional pinned
        int64 buffer (>= numPoint) so the H2D copy of the permutation is truly asynchronous."""
        lib = L.load()
        n = self.n
        perm = torch.randperm(n)[:numPoint]
        n_sel = perm.numel()
        out = torch.empty((n_sel, 2), dtype=torch.int64, device=self.cand.device)
        if n_sel:
            if staging is not None:
                staging[:n_sel].copy_(perm)
                perm = staging[:n_sel]
            perm_d = perm.to(self.cand.device, non_blocking=True)
            L.check(lib.mv_kp_gather(self.cand.data_ptr(), perm_d.data_ptr(), n_sel, self.W, out.data_ptr(), _stream()),
                    "mv_kp_gather")
        return out


_ws_cache: dict = {}


def _kp_workspace(H: int, W: int, device) -> torch.Tensor:
    # one workspace per (shape, device, STREAM): the kernel needs its counters zero on entry and leaves them zero, so two
    # launches may share a workspace only when they are ordered, i.e. on the same stream
    key = (H, W, str(device), _stream())
    ws = _ws_cache.get(key)
    if ws is None:
        nbytes = L.load().mv_kp_select_workspace_bytes(H, W)
        ws = torch.zeros((nbytes + 7) // 8, dtype=torch.int64, device=device)   # must start zeroed (see header)
        _ws_cache[key] = ws
    return ws


def kp_select(mode: str, H: int, W: int, flow_cov: torch.Tensor | None = None, depth0: torch.Tensor | None = None,
              depth0_cov: torch.Tensor | None = None, depth1: torch.Tensor | None = None,
              depth1_cov: torch.Tensor | None = None, mask_a: torch.Tensor | None = None,
              mask_b: torch.Tensor | None = None, kernel_size: int = 7, mask_width: int = 32,
              max_depth: float = 0.0, max_depth_cov: float = 0.0, max_match_cov: float = 0.0) -> KeypointCandidates:
    """Dense part of the covariance-aware selectors (KeypointSelector.py:87-97,260-327,362-400)."""
    lib = L.load()
    modes = {"nodepth": L.MV_KP_NODEPTH, "full": L.MV_KP_FULL, "mapping": L.MV_KP_MAPPING}
    dev = None
    ts = {}
    for name, t in (("flow_cov", flow_cov), ("depth0", depth0), ("depth0_cov", depth0_cov), ("depth1", depth1),
                    ("depth1_cov", depth1_cov)):
        if t is not None:
            t = _req(t, torch.float32, name)
            dev = t.device
        ts[name] = t
    ms = {}
    for name, t in (("mask_a", mask_a), ("mask_b", mask_b)):
        if t is not None:
            if t.dtype == torch.bool:
                t = t.contiguous().view(torch.uint8)
            t = _req(t, torch.uint8, name)
        ms[name] = t
    p = L.mvKpSelectParams(H, W, modes[mode], kernel_size, mask_width, max_depth, max_depth_cov, max_match_cov)
    ws = _kp_workspace(H, W, dev)
    cand = torch.empty((H * W,), dtype=torch.int32, device=dev)
    count = torch.empty((4,), dtype=torch.int32, device=dev)
    stats = torch.empty((4,), dtype=torch.float32, device=dev)
    L.check(lib.mv_kp_select(_ptr(ts["flow_cov"]), _ptr(ts["depth0"]), _ptr(ts["depth0_cov"]), _ptr(ts["depth1"]),
                             _ptr(ts["depth1_cov"]), _ptr(ms["mask_a"]), _ptr(ms["mask_b"]), C.byref(p),
                             ws.data_ptr(), ws.numel() * 8, cand.data_ptr(), count.data_ptr(), stats.data_ptr(),
                             _stream()), "mv_kp_select")
    return KeypointCandidates(cand, count, stats, W)


def pose_apply(pose: torch.Tensor, pos_Tc: torch.Tensor, cov_Tc: torch.Tensor):
    """World-frame tables from camera-frame ones and a pose (``mv_pose_apply_lanes``, MACVO.py:273-281): returns
    ``(pos_Tw [N,3] f32, rot [9] f64, cov_Tw [N,3,3] f64)`` — what ``backproject(pose=...)`` / ``match_cov(rot=...)`` produce."""
    lib = L.load()
    pose = _req(pose.reshape(7), torch.float32, "pose")
    pos_Tc = _req(pos_Tc, torch.float32, "pos_Tc")
    cov_Tc = _req(cov_Tc, torch.float64, "cov_Tc")
    n = pos_Tc.shape[0]
    dev = pos_Tc.device
    pos_Tw = torch.empty_like(pos_Tc)
    rot = torch.empty(9, dtype=torch.float64, device=dev)
    cov_Tw = torch.empty_like(cov_Tc)
    cnt = (C.c_int32 * 1)(n)
    L.check(lib.mv_pose_apply_lanes(pose.data_ptr(), pos_Tc.data_ptr(), cov_Tc.data_ptr(), 1, cnt, n, pos_Tw.data_ptr(),
                                    rot.data_ptr(), cov_Tw.data_ptr(), _stream()), "mv_pose_apply_lanes")
    return pos_Tw, rot, cov_Tw


def frontend_epilogue_select(flow: torch.Tensor, cov: torch.Tensor, baseline: float, fx: float, cov_is_log: bool = True,
                             enforce_positive_disparity: bool = False, kernel_size: int = 7, mask_width: int = 32,
                             max_match_cov: float = 0.0, mask_a: torch.Tensor | None = None,
                             mask_b: torch.Tensor | None = None) -> "tuple[FrontendMaps, KeypointCandidates]":
    """``frontend_epilogue`` followed by ``kp_select("nodepth", ...)`` in one launch less (``mv_frontend_epilogue_select_lanes``):
    the selector's first kernel derives the quality map from the network's covariance planes itself and writes the epilogue's
    maps on the way.  Bit-identical to the two calls (``test_gpu_backend::test_fused_epilogue_selector_is_bitwise_the_two_calls``)."""
    lib = L.load()
    flow = _req(flow, torch.float32, "flow")
    cov = _req(cov, torch.float32, "cov")
    assert flow.shape == cov.shape and flow.shape[0] == 2 and flow.shape[1] == 2
    _, _, H, W = flow.shape
    dev = flow.device
    mk = lambda c: torch.empty((1, c, H, W), dtype=torch.float32, device=dev)  # noqa: E731
    disparity, disparity_cov, depth, depth_cov, mflow, mcov = mk(1), mk(1), mk(1), mk(1), mk(2), mk(3)
    bad = torch.empty((1, 1, H, W), dtype=torch.bool, device=dev) if enforce_positive_disparity else None
    ms = []
    for name, t in (("mask_a", mask_a), ("mask_b", mask_b)):
        if t is not None:
            if t.dtype == torch.bool:
                t = t.contiguous().view(torch.uint8)
            t = _req(t, torch.uint8, name)
        ms.append(t)
    p = L.mvKpSelectParams(H, W, L.MV_KP_NODEPTH, kernel_size, mask_width, 0.0, 0.0, max_match_cov)
    ws = _kp_workspace(H, W, dev)
    cand = torch.empty((H * W,), dtype=torch.int32, device=dev)
    count = torch.empty((4,), dtype=torch.int32, device=dev)
    stats = torch.empty((4,), dtype=torch.float32, device=dev)
    bl_fx = float(baseline) * float(fx)
    L.check(lib.mv_frontend_epilogue_select_lanes(flow.data_ptr(), cov.data_ptr(), int(cov_is_log), bl_fx, bl_fx ** 2,
                                                  disparity.data_ptr(), disparity_cov.data_ptr(), depth.data_ptr(),
                                                  depth_cov.data_ptr(), _ptr(bad), mflow.data_ptr(), mcov.data_ptr(),
                                                  _ptr(ms[0]), _ptr(ms[1]), C.byref(p), ws.data_ptr(), ws.numel() * 8,
                                                  cand.data_ptr(), count.data_ptr(), stats.data_ptr(), 1, _stream()),
            "mv_frontend_epilogue_select_lanes")
    return FrontendMaps(depth, depth_cov, disparity, disparity_cov, bad, mflow, mcov), KeypointCandidates(cand, count, stats, W)


# ------------------------------------------------------------------------------------------- A12
@dataclass
class TrackedKeypoints:
    kp0_uv: torch.Tensor      # [N,2] float32 (kp0 as float)
    kp1_uv: torch.Tensor      # [N,2] float32
    inbound: torch.Tensor     # [N] bool
    vals: torch.Tensor        # [11,N] float32 rows: d0 disp0 sdisp0 sdd0 d1 disp1 sdisp1 sdd1 suu svv suv
    sigma0: torch.Tensor      # [N,3] float32 default sigma of kp0
    sigma1: torch.Tensor      # [N,3] float32 match covariance (read at kp0)


def kp_track(kp0_uv: torch.Tensor, flow: torch.Tensor, flow_cov: torch.Tensor | None, depth0: FrontendMaps | dict,
             depth1: FrontendMaps | dict, edge: int, match_cov_default: float = 0.25) -> TrackedKeypoints:
    """kp1 = kp0 + flow[kp0], strict border test, and all per-keypoint gathers (MACVO.py:198-232) in one launch."""
    lib = L.load()
    kp0_uv = _req(kp0_uv, torch.int64, "kp0_uv")
    flow = _req(flow, torch.float32, "flow")
    _, _, H, W = flow.shape
    N = kp0_uv.shape[0]
    dev = flow.device

    def g(d, k):
        v = d[k] if isinstance(d, dict) else getattr(d, k)
        return None if v is None else _req(v, torch.float32, k)

    d0 = [g(depth0, k) for k in ("depth", "disparity", "disparity_cov", "depth_cov")]
    d1 = [g(depth1, k) for k in ("depth", "disparity", "disparity_cov", "depth_cov")]
    fc = None if flow_cov is None else _req(flow_cov, torch.float32, "flow_cov")
    kp0f = torch.empty((N, 2), dtype=torch.float32, device=dev)
    kp1 = torch.empty((N, 2), dtype=torch.float32, device=dev)
    inb = torch.empty((N,), dtype=torch.bool, device=dev)
    vals = torch.empty((11, N), dtype=torch.float32, device=dev)
    s0 = torch.empty((N, 3), dtype=torch.float32, device=dev)
    s1 = torch.empty((N, 3), dtype=torch.float32, device=dev)
    L.check(lib.mv_kp_track(kp0_uv.data_ptr(), N, flow.data_ptr(), _ptr(fc), *[_ptr(t) for t in d0],
                            *[_ptr(t) for t in d1], H, W, edge, match_cov_default, kp0f.data_ptr(), kp1.data_ptr(),
                            inb.data_ptr(), vals.data_ptr(), s0.data_ptr(), s1.data_ptr(), _stream()), "mv_kp_track")
    return TrackedKeypoints(kp0f, kp1, inb, vals, s0, s1)


# ------------------------------------------------------------------------------------------- A13-A16
def match_cov(depth_map: torch.Tensor, kp_uv: torch.Tensor, flow_cov: torch.Tensor, depth_cov: torch.Tensor | None,
              fx: float, fy: float, cx: float, cy: float, kernel_size: int = 31, min_flow_cov: float = 0.25,
              min_depth_cov: float = 0.05, use_patch_var: bool = True, rot: torch.Tensor | None = None,
              want_stats: bool = False):
    """MAC-VO covariance model (Project2to3.py:124-181,377-433; Math.py:43-63).  ``flow_cov [N,3]`` is clamped IN
    PLACE like the reference.  Returns ``cov [N,3,3] float64`` (GPU) and, if ``rot [3,3] float64`` is given,
    ``R cov R^T`` (MACVO.py:273-281)."""
    lib = L.load()
    depth_map = _req(depth_map, torch.float32, "depth_map")
    H, W = depth_map.shape[-2:]
    if kp_uv.dtype != torch.float32:
        kp_uv = kp_uv.to(torch.float32)
    kp_uv = _req(kp_uv, torch.float32, "kp_uv")
    if flow_cov.dtype != torch.float32 or not flow_cov.is_contiguous() or not flow_cov.is_cuda:
        raise L.MacvoHipError("match_cov: flow_cov must be a contiguous float32 GPU tensor (it is clamped in place)")
    N = kp_uv.shape[0]
    dev = depth_map.device
    dc = None if depth_cov is None else _req(depth_cov, torch.float32, "depth_cov")
    r = None if rot is None else _req(rot.to(dev), torch.float64, "rot")
    out = torch.empty((N, 3, 3), dtype=torch.float64, device=dev)
    out_rot = torch.empty((N, 3, 3), dtype=torch.float64, device=dev) if r is not None else None
    stats = torch.empty((N, 2), dtype=torch.float32, device=dev) if want_stats else None
    p = L.mvMatchCovParams(H, W, kernel_size, int(use_patch_var or dc is None), fx, fy, cx, cy, min_flow_cov ** 2,
                           min_depth_cov)
    L.check(lib.mv_match_cov(depth_map.data_ptr(), kp_uv.data_ptr(), flow_cov.data_ptr(), _ptr(dc), _ptr(r),
                             C.byref(p), N, out.data_ptr(), _ptr(out_rot), _ptr(stats), _stream()), "mv_match_cov")
    res = (out,)
    if out_rot is not None:
        res += (out_rot,)
    if want_stats:
        res += (stats,)
    return res[0] if len(res) == 1 else res


def match_cov_pair(depth0: torch.Tensor, kp0_uv: torch.Tensor, sigma0: torch.Tensor, depth1: torch.Tensor,
                   kp1_uv: torch.Tensor, sigma1: torch.Tensor, fx: float, fy: float, cx: float, cy: float,
                   rot: torch.Tensor | None = None, kernel_size: int = 31, min_flow_cov: float = 0.25,
                   min_depth_cov: float = 0.05):
    """Both covariance-model calls of a frame (MACVO.py:241-242) in one launch -> (cov0, cov0_world | None, cov1)."""
    lib = L.load()
    d0, d1 = _req(depth0, torch.float32, "depth0"), _req(depth1, torch.float32, "depth1")
    H, W = d0.shape[-2:]
    k0, k1 = _req(kp0_uv, torch.float32, "kp0_uv"), _req(kp1_uv, torch.float32, "kp1_uv")
    for t_, nm in ((sigma0, "sigma0"), (sigma1, "sigma1")):
        if t_.dtype != torch.float32 or not t_.is_contiguous() or not t_.is_cuda:
            raise L.MacvoHipError(f"match_cov_pair: {nm} must be a contiguous float32 GPU tensor (clamped in place)")
    N, dev = k0.shape[0], d0.device
    assert k1.shape[0] == N
    r = None if rot is None else _req(rot.to(dev), torch.float64, "rot")
    c0 = torch.empty((N, 3, 3), dtype=torch.float64, device=dev)
    c1 = torch.empty((N, 3, 3), dtype=torch.float64, device=dev)
    c0w = torch.empty((N, 3, 3), dtype=torch.float64, device=dev) if r is not None else None
    p = L.mvMatchCovParams(H, W, kernel_size, 1, fx, fy, cx, cy, min_flow_cov ** 2, min_depth_cov)
    L.check(lib.mv_match_cov_pair(d0.data_ptr(), k0.data_ptr(), sigma0.data_ptr(), _ptr(r), c0.data_ptr(), _ptr(c0w),
                                  d1.data_ptr(), k1.data_ptr(), sigma1.data_ptr(), c1.data_ptr(), C.byref(p), N, _stream()),
            "mv_match_cov_pair")
    return c0, c0w, c1


# ------------------------------------------------------------------------------------------- A17-A22
def lm_default_params() -> L.mvLMParams:
    p = L.mvLMParams()
    L.load().mv_lm_default_params(C.byref(p))
    return p


@dataclass
class PGOBatch:
    """Concatenated per-point arrays of ``nprob`` independent two-frame problems (all GPU tensors)."""
    offsets: torch.Tensor            # [nprob+1] int32
    init_pose: torch.Tensor          # [nprob,7] float32
    intrinsics: torch.Tensor         # [nprob,4] float32 (fx fy cx cy)
    baseline: torch.Tensor           # [nprob] float32
    pos_Tw: torch.Tensor             # [Ntot,3] float32
    pixel2_uv: torch.Tensor          # [Ntot,2] float32
    cov_Tw: torch.Tensor | None = None           # [Ntot,3,3] float64 (icp)
    pixel2_d: torch.Tensor | None = None         # [Ntot] float32 (icp)
    pixel2_disp: torch.Tensor | None = None      # [Ntot] float32 (disp)
    pixel2_disp_cov: torch.Tensor | None = None  # [Ntot] float32 (disp)
    pixel2_uv_cov: torch.Tensor | None = None    # [Ntot,3] float32 (reproj/disp)
    obs2_covTc: torch.Tensor | None = None       # [Ntot,3,3] float64 (icp)
    valid: torch.Tensor | None = None            # [Ntot] bool/uint8: rows to use (None = all)


_GRAPH = {"icp": L.MV_GRAPH_ICP, "reproj": L.MV_GRAPH_REPROJ, "disp": L.MV_GRAPH_DISP}


def pgo_solve(batch: PGOBatch, graph_type: str = "disp", params: L.mvLMParams | None = None, min_points: int = 0,
              out_pose_f32: torch.Tensor | None = None):
    """Batched two-frame PGO (Optimizer.py:81-102 + PyposeOptimizers.py:160-194) -> (pose [nprob,7] f64, info [nprob,4] f64).
    ``out_pose_f32`` (optional ``[nprob,7]`` float32 GPU tensor) receives ``motion.float()`` (Optimizer.py:104-108)."""
    lib = L.load()
    p = params or lm_default_params()
    nprob = batch.init_pose.shape[0]
    dev = batch.init_pose.device
    f32 = lambda t, n: None if t is None else _req(t, torch.float32, n)  # noqa: E731
    f64 = lambda t, n: None if t is None else _req(t, torch.float64, n)  # noqa: E731
    out_pose = torch.empty((nprob, 7), dtype=torch.float64, device=dev)
    out_info = torch.empty((nprob, 4), dtype=torch.float64, device=dev)
    L.check(lib.mv_pgo_solve(nprob, _req(batch.offsets, torch.int32, "offsets").data_ptr(), _GRAPH[graph_type],
                             f32(batch.init_pose, "init_pose").data_ptr(), f32(batch.intrinsics, "intrinsics").data_ptr(),
                             f32(batch.baseline, "baseline").data_ptr(), f32(batch.pos_Tw, "pos_Tw").data_ptr(),
                             _ptr(f64(batch.cov_Tw, "cov_Tw")), f32(batch.pixel2_uv, "pixel2_uv").data_ptr(),
                             _ptr(f32(batch.pixel2_d, "pixel2_d")), _ptr(f32(batch.pixel2_disp, "pixel2_disp")),
                             _ptr(f32(batch.pixel2_disp_cov, "pixel2_disp_cov")),
                             _ptr(f32(batch.pixel2_uv_cov, "pixel2_uv_cov")), _ptr(f64(batch.obs2_covTc, "obs2_covTc")),
                             _ptr(_u8(batch.valid)), int(min_points), C.byref(p), out_pose.data_ptr(),
                             out_info.data_ptr(), _ptr(out_pose_f32), _stream()), "mv_pgo_solve")
    return out_pose, out_info


# ------------------------------------------------------------------------------------------- A12 tail / X1
def backproject(kp_uv: torch.Tensor, depth_vals: torch.Tensor, K4: tuple, pose: torch.Tensor | None,
                want_rot: bool = False):
    """``pixel2point_NED`` (Point.py:15-17) + ``prev_pose.Act`` + ``rotation().matrix().double()`` (MACVO.py:240,273-281).
    ``depth_vals`` may be a strided column view (e.g. ``tracked.vals[:, 0]``).  Returns (pos_Tc, pos_Tw | None, rot | None)."""
    lib = L.load()
    if kp_uv.dtype != torch.float32:
        kp_uv = kp_uv.to(torch.float32)
    kp_uv = _req(kp_uv, torch.float32, "kp_uv")
    assert depth_vals.dtype == torch.float32 and depth_vals.is_cuda and depth_vals.dim() == 1
    stride = depth_vals.stride(0) if depth_vals.numel() > 1 else 1
    N = kp_uv.shape[0]
    dev = kp_uv.device
    pos_Tc = torch.empty((N, 3), dtype=torch.float32, device=dev)
    pos_Tw = torch.empty((N, 3), dtype=torch.float32, device=dev) if pose is not None else None
    rot = torch.empty((3, 3), dtype=torch.float64, device=dev) if (want_rot and pose is not None) else None
    pose_c = None if pose is None else _req(pose.reshape(-1), torch.float32, "pose")
    L.check(lib.mv_backproject(kp_uv.data_ptr(), depth_vals.data_ptr(), int(stride), *[float(k) for k in K4],
                               _ptr(pose_c), N, pos_Tc.data_ptr(), _ptr(pos_Tw), _ptr(rot), _stream()), "mv_backproject")
    return pos_Tc, pos_Tw, rot


@dataclass
class MapPoints:
    """Dense map points of one frame (MACVO.py:313-337): camera-frame covariance is UNROTATED, as the reference stores it."""
    uv: torch.Tensor            # [N,2] float32 (integer-valued pixel coordinates)
    depth: torch.Tensor         # [N]
    sigma_dd: torch.Tensor      # [N]  depth variance at the pixel
    pos_Tc: torch.Tensor        # [N,3]
    pos_Tw: torch.Tensor        # [N,3]
    cov_Tc: torch.Tensor        # [N,3,3] float64
    color: torch.Tensor | None  # [N,3] uint8


def map_points(uv: torch.Tensor, depth: torch.Tensor, depth_cov: torch.Tensor, K4: tuple, pose: torch.Tensor,
               image: torch.Tensor | None = None, match_cov_default: float = 0.25, kernel_size: int = 31,
               min_flow_cov: float = 0.25, min_depth_cov: float = 0.05) -> MapPoints:
    """Everything ``run_pair`` does for the dense map after the selector (MACVO.py:317-334) in two launches."""
    lib = L.load()
    uv = _req(uv, torch.int64, "uv")
    depth = _req(depth, torch.float32, "depth")
    depth_cov = _req(depth_cov, torch.float32, "depth_cov")
    H, W = depth.shape[-2:]
    N, dev = uv.shape[0], depth.device
    img = None if image is None else _req(image.reshape(3, H, W), torch.float32, "image")
    f = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)  # noqa: E731
    uvf, d, sdd, sig, Tc, Tw = f(N, 2), f(N), f(N), f(N, 3), f(N, 3), f(N, 3)
    col = torch.empty((N, 3), dtype=torch.uint8, device=dev) if img is not None else None
    pose_c = _req(pose.reshape(-1), torch.float32, "pose")
    L.check(lib.mv_map_points(uv.data_ptr(), N, depth.data_ptr(), depth_cov.data_ptr(), _ptr(img), H, W,
                              *[float(k) for k in K4], pose_c.data_ptr(), float(match_cov_default), uvf.data_ptr(),
                              d.data_ptr(), sdd.data_ptr(), sig.data_ptr(), Tc.data_ptr(), Tw.data_ptr(), _ptr(col), _stream()),
            "mv_map_points")
    cov = match_cov(depth, uvf, sig, sdd, *K4, kernel_size=kernel_size, min_flow_cov=min_flow_cov, min_depth_cov=min_depth_cov)
    return MapPoints(uvf, d, sdd, Tc, Tw, cov, col)


FILTER_COV_SANITY, FILTER_SIMPLE_DEPTH, FILTER_FRONT_OF_CAM = 1, 2, 4


def obs_filter(inbound: torch.Tensor | None, cov1: torch.Tensor | None, cov2: torch.Tensor | None,
               vals: torch.Tensor | None, flags: int = FILTER_COV_SANITY, min_depth: float = 0.0,
               max_depth: float = 0.0):
    """Fused observation filters (OutlierFilter.py:91-137) -> (valid [N] bool, count [1] int32), both on the GPU."""
    lib = L.load()
    sizes = [t.shape[0] for t in (inbound, cov1, cov2) if t is not None] + ([vals.shape[1]] if vals is not None else [])
    if not sizes or any(n != sizes[0] for n in sizes):
        raise L.MacvoHipError(f"obs_filter: inbound [N], cov1/cov2 [N,3,3] and vals [11,N] must agree on N (got {sizes})")
    N = sizes[0]
    dev = next(t for t in (inbound, cov1, cov2, vals) if t is not None).device
    valid = torch.empty((N,), dtype=torch.bool, device=dev)
    count = torch.empty((1,), dtype=torch.int32, device=dev)
    c1 = None if cov1 is None else _req(cov1, torch.float64, "cov1")
    c2 = None if cov2 is None else _req(cov2, torch.float64, "cov2")
    vv = None if vals is None else _req(vals, torch.float32, "vals")
    L.check(lib.mv_obs_filter(_ptr(_u8(inbound)), _ptr(c1), _ptr(c2), _ptr(vv), flags, min_depth, max_depth, N,
                              valid.data_ptr(), count.data_ptr(), _stream()), "mv_obs_filter")
    return valid, count
