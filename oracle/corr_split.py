"""CPU restatement of the SPLIT cost-volume arithmetic (TEST INFRASTRUCTURE — nothing under mac-vo_amd/ imports this).

The reference computes ``corr = einsum('bhid,bhjd->bhij')`` (``Module/Network/FlowFormerCov/flownet.py:26``; FlowFormer
``MemoryEncoder.corr``, source in the empty submodule: parity unpinned, see oracle/corr.py) in TF32 / fp16 on its own hardware
(``Module/Frontend/Frontend.py:275-277``, ``MACVO_Fast.yaml:69-76``).  The HIP path's `f16x2` / `bf16x3` modes keep the fp32
parity bar instead; this file restates what they compute so that the bar can be checked WITHOUT a GPU:

* ``pack_f16x2``  every row (one pixel's feature vector) is scaled by the power of two that puts its largest magnitude into
  [2^14, 2^15) (any finite non-zero row of the fp32 range: the shift is not clamped), then split into two fp16 pieces by successive rounding: x * 2^sh = h0 + h1 + O(2^-22 |x * 2^sh|);
* ``pack_bf16x3`` three bf16 pieces by successive rounding (no scaling: bf16 has fp32's exponent range);
* ``corr_volume_split``  the piece products (h0 h0 + h0 h1 + h1 h0, or the six bf16 products with i + j <= 2), every product exact in
  fp32 arithmetic terms (11 x 11 / 8 x 8 bits), accumulated in fp32, the row scales undone exactly afterwards.

The accumulation ORDER of the MFMA units is not restated (it is not observable at the tolerance in question), so this is a
yardstick for the error bound, not a bit-level twin of the kernel.
"""
from __future__ import annotations

import numpy as np


def pack_f16x2(f: np.ndarray):
    """f [N, C] float32 -> (h0, h1 float16 [N, C], sh int32 [N]) with f * 2^sh ~= h0 + h1."""
    f = np.asarray(f, dtype=np.float32)
    m = np.abs(f).max(axis=1)
    sh = np.zeros(f.shape[0], dtype=np.int32)
    ok = (m > 0) & np.isfinite(m)
    sh[ok] = 14 - np.floor(np.log2(m[ok].astype(np.float64))).astype(np.int32)
    x = np.ldexp(f, sh[:, None]).astype(np.float32)
    h0 = x.astype(np.float16)
    h1 = (x - h0.astype(np.float32)).astype(np.float16)
    return h0, h1, sh


def _to_bf16(x: np.ndarray) -> np.ndarray:
    """round-to-nearest-even float32 -> bfloat16, returned as float32 values"""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def pack_bf16x3(f: np.ndarray):
    f = np.asarray(f, dtype=np.float32)
    p0 = _to_bf16(f)
    r1 = f - p0
    p1 = _to_bf16(r1)
    p2 = _to_bf16(r1 - p1)
    return p0, p1, p2


def corr_volume_split(f1: np.ndarray, f2: np.ndarray, mode: str = "f16x2") -> np.ndarray:
    """f1 [N1, C], f2 [N2, C] float32 -> [N1, N2] float32 as the split kernels compute it (up to accumulation order)."""
    if mode == "f16x2":
        a0, a1, sa = pack_f16x2(f1)
        b0, b1, sb = pack_f16x2(f2)
        A0, A1, B0, B1 = (t.astype(np.float32) for t in (a0, a1, b0, b1))
        acc = (A0 @ B1.T).astype(np.float32)           # smallest first
        acc = (acc + (A1 @ B0.T).astype(np.float32)).astype(np.float32)
        acc = (acc + (A0 @ B0.T).astype(np.float32)).astype(np.float32)
        return np.ldexp(acc, -(sa[:, None] + sb[None, :])).astype(np.float32)
    if mode == "bf16x3":
        a = pack_bf16x3(f1)
        b = pack_bf16x3(f2)
        acc = np.zeros((f1.shape[0], f2.shape[0]), dtype=np.float32)
        for i, j in ((0, 2), (1, 1), (2, 0), (0, 1), (1, 0), (0, 0)):
            acc = (acc + (a[i] @ b[j].T).astype(np.float32)).astype(np.float32)
        return acc
    raise ValueError(mode)
