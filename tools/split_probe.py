"""Timing of the split + streaming volume (pack, GEMM, both) next to the exact fp32 kernel, HIP events, GPU to itself; and the
largest deviation of each precision from the fp64 product (sampled rows) in units of the parity bar 2e-5 sqrt(C)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from macvo_amd import ops

def t(fn, n=100, warm=30):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

shapes = [(2, 60, 80), (2, 90, 160), (64, 60, 80)] if len(sys.argv) < 2 else [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for B, H, W in shapes:
    C, N = 256, H * W
    g = torch.Generator().manual_seed(0)
    f1 = torch.randn(B, C, H, W, generator=g).cuda(); f2 = torch.randn(B, C, H, W, generator=g).cuda()
    out = torch.empty((B * N, 1, H, W), device="cuda")
    n = 100 if B * N * N < 1e9 else 12
    for _ in range(3 * n): ops.corr_volume(f1, f2, out=out)            # clocks
    fl = 2.0 * B * N * N * C
    for mode, nprod in (("bf16x3", 6), ("f16x2", 3)):
        pk = ops.volume_pack(f1, f2, mode=mode)
        t_pack = t(lambda: ops.volume_pack(f1, f2, out=pk, mode=mode), n, n // 3)
        t_gemm = t(lambda: ops.corr_volume_packed(pk[0], pk[1], B, C, N, N, out=out, mode=mode), n, n // 3)
        print(f"B={B} {H}x{W} {mode}: pack {t_pack:.1f} us | split GEMM {t_gemm:.1f} us = {fl / t_gemm / 1e6:.1f} TF algorithmic "
              f"({nprod * fl / t_gemm / 1e6 / 2500:.3f} of 2.5 PF executed; write {B * N * N * 4 / t_gemm / 1e3:.0f} GB/s)", flush=True)
    t_exact = t(lambda: ops.corr_volume(f1, f2, out=out), n, n // 3)
    print(f"B={B} {H}x{W} exact fp32 {t_exact:.1f} us = {fl / t_exact / 1e6 / 157.3:.3f} of 157.3 TF", flush=True)
    if B * N * N < 1e9:
        rows = torch.arange(0, N, 37)
        for b in range(B):
            a64, b64 = f1[b].reshape(C, N).double(), f2[b].reshape(C, N).double()
            ref = a64[:, rows.cuda()].T @ b64
            for prec in ("exact", "bf16x3", "f16x2"):
                v = ops.corr_volume(f1, f2, precision=prec)[b * N:(b + 1) * N].reshape(N, N)[rows.cuda()].double()
                e = (v - ref).abs().max().item()
                print(f"    pair {b} {prec:7s} max |out - fp64| = {e:.3e} = {e / (2e-5 * C ** 0.5):.3f} of the bar (N(0,1) features)", flush=True)
