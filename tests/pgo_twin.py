"""TEST INFRASTRUCTURE: builds and calls tests/c_abi/pgo_twin.cpp, the host replay of the PGO kernel.

``pgo_twin.cpp`` includes the kernel's own arithmetic header (``mac-vo_amd/csrc/pgo_math.h``) and replays the kernel's loop
and reduction trees lane by lane with g++.  The CPU suite pins it to the oracle and to the reference golden, the GPU suite pins
the kernel to it.  The product path never builds or loads it."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SRC = os.path.join(ROOT, "tests", "c_abi", "pgo_twin.cpp")
_HDR = os.path.join(ROOT, "mac-vo_amd", "csrc", "pgo_math.h")
_lib = None


def build() -> C.CDLL:
    """g++ -O2 -ffp-contract=off (fma() stays the one fused operation, nothing else is contracted) into a scratch directory."""
    global _lib
    if _lib is not None:
        return _lib
    out_dir = os.path.join(tempfile.gettempdir(), f"macvo_pgo_twin_{os.getuid()}")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libpgo_twin.so")
    newest = max(os.path.getmtime(_SRC), os.path.getmtime(_HDR))
    if not os.path.exists(so) or os.path.getmtime(so) < newest:
        tmp = so + f".{os.getpid()}.tmp"
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                        _SRC, "-o", tmp, "-lm"], check=True)
        os.replace(tmp, so)
    _lib = C.CDLL(so)
    _lib.pgo_twin_solve.restype = C.c_int
    return _lib


def solve(batch, graph_type: str = "disp", params=None, min_points: int = 0, nw: int = 0, spec: int = 1):
    """Same contract as ``ops.pgo_solve`` on a CPU ``PGOBatch``: (pose [nprob,7] f64, info [nprob,4] f64)."""
    from macvo_amd import _lib as L
    from macvo_amd import ops

    lib = build()
    p = params or ops.lm_default_params()
    nprob = batch.init_pose.shape[0]

    keep = []

    def ptr(t, dt):
        if t is None:
            return C.c_void_p(None)
        t = t.detach().to("cpu", dt).contiguous()
        keep.append(t)
        return C.c_void_p(t.data_ptr())

    valid = None if batch.valid is None else batch.valid.to(torch.uint8)
    out_pose = torch.zeros((nprob, 7), dtype=torch.float64)
    out_info = torch.zeros((nprob, 4), dtype=torch.float64)
    gt = {"icp": L.MV_GRAPH_ICP, "reproj": L.MV_GRAPH_REPROJ, "disp": L.MV_GRAPH_DISP}[graph_type]
    rc = lib.pgo_twin_solve(
        C.c_int(nprob), ptr(batch.offsets, torch.int32), C.c_int(gt), ptr(batch.init_pose, torch.float32),
        ptr(batch.intrinsics, torch.float32), ptr(batch.baseline, torch.float32), ptr(batch.pos_Tw, torch.float32),
        ptr(batch.cov_Tw, torch.float64), ptr(batch.pixel2_uv, torch.float32), ptr(batch.pixel2_d, torch.float32),
        ptr(batch.pixel2_disp, torch.float32), ptr(batch.pixel2_disp_cov, torch.float32), ptr(batch.pixel2_uv_cov, torch.float32),
        ptr(batch.obs2_covTc, torch.float64), ptr(valid, torch.uint8), C.c_int(int(min_points)), C.byref(p),
        C.c_void_p(out_pose.data_ptr()), C.c_void_p(out_info.data_ptr()), C.c_void_p(None), C.c_int(nw), C.c_int(spec))
    assert rc == 0, rc
    return out_pose, out_info
