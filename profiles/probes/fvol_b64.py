import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from macvo_amd import ops
B, C, H, W = 64, 256, 60, 80
g = torch.Generator(device="cuda").manual_seed(3)
f1 = torch.randn(B, C, H, W, device="cuda", generator=g); f2 = torch.randn(B, C, H, W, device="cuda", generator=g)
out = torch.empty(B * H * W, 1, H, W, device="cuda")
for _ in range(10): ops.corr_volume(f1, f2, "chw", out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.corr_volume(f1, f2, "chw", out=out)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
fl = 2.0 * B * (H * W) ** 2 * C
print(f"MV_VOL_DMA={os.environ.get('MV_VOL_DMA','(32)')} B=64: {us:.0f} us  {fl / us / 1e6:.1f} TF/s ({fl / us / 1e6 / 157.3:.3f})")
