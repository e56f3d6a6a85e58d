"""Probe (round 6): the B = 64 lookup on fp32 cells, row-major vs 4 x 4 tiles (mv_volume_pack_tiled + mv_corr_lookup_tiled), with the pair-form tap phase."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from macvo_amd import ops
from tools.synth import coords_grid

dev = torch.device("cuda:0")
B, C, H, W = 64, 256, 60, 80
N = H * W
g = torch.Generator().manual_seed(1)
f1 = torch.randn(B, C, H, W, generator=g).to(dev); f2 = torch.randn(B, C, H, W, generator=g).to(dev)
coords = (coords_grid(B, H, W) + (torch.rand(B, 2, H, W, generator=g) * 2 - 1) * 8).to(dev)
tok = torch.empty(B, 81, H, W, device=dev)
pk = ops.volume_pack(f1, f2, mode="f16x2")
vol = ops.corr_volume_packed(pk[0], pk[1], B, C, N, N, mode="f16x2").view(B * N, 1, H, W)
ref = ops.corr_lookup(vol, coords, 4).clone()


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


print("row-major fp32 B=64: %.1f us" % timeit(lambda: ops.corr_lookup(vol, coords, 4, out=tok)))
del vol
pkt = ops.volume_pack(f1, f2, mode="f16x2", tiled_hw=(H, W))
volt = ops.corr_volume_packed(pkt[0], pkt[1], B, C, N, N, mode="f16x2").view(B * N, 1, H, W)
assert torch.equal(ops.corr_lookup(volt, coords, 4, tiled=True), ref)
print("tiled fp32     B=64: %.1f us" % timeit(lambda: ops.corr_lookup(volt, coords, 4, out=tok, tiled=True)))
