"""TEST INFRASTRUCTURE: the phase-by-phase cost patch-embedding kernel as a stand-alone library.

Until round 5 ``mac-vo_amd/csrc/patch_embed.hip`` shipped two whole-slice kernels for 60 / 64 x 80 slices: the phase-by-phase one and the pipelined one of
``patch_embed_v3.hip`` (same 16-bit values meeting in the same k order per output: bit-identical tokens).  The product library now carries only the pipelined
kernel; the phase kernel is compiled from the same source with ``-DMV_PE_PHASE_REFERENCE`` into a test-only library (built here with hipcc on first use, cached in the
temp directory) and stays the BITWISE reference of the pipelined kernel (tests/test_gpu_patch_embed.py)."""
import ctypes as C
import os
import subprocess
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SRC = os.path.join(ROOT, "mac-vo_amd", "csrc", "patch_embed.hip")
_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    out_dir = os.path.join(tempfile.gettempdir(), f"macvo_pe_phase_ref_{os.getuid()}")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libmacvo_pe_phase_ref.so")
    deps = [_SRC] + [os.path.join(ROOT, "mac-vo_amd", "csrc", h) for h in ("patch_embed_dev.h", "common.h")] + [os.path.join(ROOT, "include", "macvo_hip.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        cmd = [os.environ.get("HIPCC", "hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
               "-Wno-unused-function", "-DMV_PE_PHASE_REFERENCE", "-shared", "-o", so + ".tmp", _SRC]
        subprocess.run(cmd, check=True, capture_output=True, text=True)
        os.replace(so + ".tmp", so)
    _lib = C.CDLL(so)
    P = C.c_void_p
    _lib.mv_cost_patch_embed_phase_ref.restype = C.c_int
    _lib.mv_cost_patch_embed_phase_ref.argtypes = [P, C.c_int, P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P]
    return _lib


def cost_patch_embed(x: torch.Tensor, packed, tokens: bool = False, out_dtype=None) -> torch.Tensor:
    """Same contract as ``ops.cost_patch_embed`` for 60 / 64 x 80 slices, through the phase-by-phase reference kernel.  ``packed``: an ``ops.PatchEmbedWeights``
    (its packed fragment buffer is shared with the product library: the layout is the same)."""
    from macvo_amd import _lib as L

    lib = _load()
    S, _, H2, W2 = x.shape
    code = {torch.float32: L.MV_F32, torch.float16: L.MV_F16, torch.bfloat16: L.MV_BF16}
    odt = out_dtype if out_dtype is not None else x.dtype
    h, w = (H2 + 7) // 8, (W2 + 7) // 8
    out = torch.empty((S, h * w, 64) if tokens else (S, 64, h, w), dtype=odt, device=x.device)
    x = x.contiguous()
    rc = lib.mv_cost_patch_embed_phase_ref(x.data_ptr(), code[x.dtype], packed.packed.data_ptr(), out.data_ptr(), code[odt], S, H2, W2, int(tokens),
                                           L.MV_F16 if packed.operand == "f16" else L.MV_BF16, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    return out
