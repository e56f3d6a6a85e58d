"""Scratch: what do the vendor GEMM libraries reach on the cost-volume shape (fp32, 2 x [4800 x 256] x [256 x 4800])?"""
import torch
dev = "cuda"
B, N, C = 2, 4800, 256
a = torch.randn(B, N, C, device=dev)
b = torch.randn(B, N, C, device=dev)
out = torch.empty(B, N, N, device=dev)
bt = b.transpose(1, 2).contiguous()
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
fl = B * 2.0 * N * N * C
for name, f in (("bmm a @ b^T (NT)", lambda: torch.bmm(a, b.transpose(1, 2), out=out)),
                ("bmm a @ bt  (NN)", lambda: torch.bmm(a, bt, out=out)),
                ("einsum (reference formulation)", lambda: torch.einsum("bid,bjd->bij", a, b))):
    torch.backends.cuda.matmul.allow_tf32 = False
    us = timeit(f)
    print(f"{name:32s} fp32      {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")
torch.backends.cuda.matmul.allow_tf32 = True
us = timeit(lambda: torch.bmm(a, b.transpose(1, 2), out=out))
print(f"{'bmm NT allow_tf32':32s}           {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")
ah, bh = a.half(), b.half()
us = timeit(lambda: torch.bmm(ah, bh.transpose(1, 2)))
print(f"{'bmm NT fp16 in/out':32s}           {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")
from macvo_amd import ops
f1 = a.permute(0, 2, 1).contiguous().view(B, C, 60, 80); f2 = b.permute(0, 2, 1).contiguous().view(B, C, 60, 80)
vol = torch.empty(B * N, 1, 60, 80, device=dev)
us = timeit(lambda: ops.corr_volume(f1, f2, "chw", out=vol))
print(f"{'mv_corr_volume f32 chw':32s}           {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")
