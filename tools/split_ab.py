#!/usr/bin/env python
"""A/B probe of the split volume GEMM: one library (MACVO_HIP_LIB) per process.  Checks the result against a float64 einsum on sampled rows
(so a variant that computes garbage cannot win), then times back-to-back launches between one HIP-event pair (clocks stay up), three rounds.

    MACVO_HIP_LIB=profiles/probes/libmacvo_hip_r4_both.so python tools/split_ab.py [--mode f16x2] [--B 2] [--zeros]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from macvo_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="f16x2")
ap.add_argument("--B", type=int, default=2)
ap.add_argument("--H", type=int, default=60)
ap.add_argument("--W", type=int, default=80)
ap.add_argument("--launches", type=int, default=300)
ap.add_argument("--zeros", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
C, n = 256, a.H * a.W
g = torch.Generator().manual_seed(0)
f1, f2 = torch.randn(a.B, C, a.H, a.W, generator=g), torch.randn(a.B, C, a.H, a.W, generator=g)
d1, d2 = f1.to(dev), f2.to(dev)
vol = torch.empty((a.B * n, 1, a.H, a.W), dtype=torch.float32, device=dev)
pk = ops.volume_pack(d1, d2, mode=a.mode)
ops.corr_volume_packed(pk[0], pk[1], a.B, C, n, n, out=vol, mode=a.mode)
kern = ops.last_volume_kernel()
rows = torch.randint(0, a.B * n, (256,), generator=g)
got = vol.view(a.B * n, n)[rows.to(dev)].cpu().double()
bb, ii = rows // n, rows % n
ref = torch.einsum("rc,rcn->rn", f1.reshape(a.B, C, n).double()[bb, :, ii], f2.reshape(a.B, C, n).double()[bb])
err = float((got - ref).abs().max())
ok = err <= 2e-5 * C ** 0.5
if a.zeros:
    pk = ops.volume_pack(torch.zeros_like(d1), torch.zeros_like(d2), mode=a.mode)
launch = lambda: ops.corr_volume_packed(pk[0], pk[1], a.B, C, n, n, out=vol, mode=a.mode)  # noqa: E731
for _ in range(a.launches):
    launch()
ts = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.launches):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3 / a.launches)
print(f"{os.path.basename(os.environ.get('MACVO_HIP_LIB', 'libmacvo_hip.so')):32s} {kern:34s} B={a.B} {a.H}x{a.W} {'zeros ' if a.zeros else ''}"
      f"us/launch {min(ts):7.2f} (rounds {', '.join(f'{t:.2f}' for t in ts)})  max|err| {err:.2e} {'OK' if ok else 'WRONG'}", flush=True)
