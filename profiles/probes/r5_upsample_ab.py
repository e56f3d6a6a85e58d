"""Round 5: convex_upsample A/B on the GPU box — back-to-back launches between two HIP events (no per-launch host gaps in the figure).
    MV_UPS_NSX=4|8 python profiles/probes/r5_upsample_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from macvo_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (B, h, w) in ((2, 60, 80), (2, 90, 160)):
    fl = torch.randn(B, 2, h, w, generator=g).to(dev)
    mk32 = torch.randn(B, 576, h, w, generator=g).to(dev)
    for name, mk in (("f32", mk32), ("bf16", mk32.bfloat16()), ("f16", mk32.half())):
        for _ in range(5):
            ops.convex_upsample(fl, mk, 0.25)
        torch.cuda.synchronize()
        n = 200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ops.convex_upsample(fl, mk, 0.25)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        byts = B * h * w * (576 * mk.element_size() + 2 * 4 + 128 * 4)
        print(f"NSX={os.environ.get('MV_UPS_NSX','auto')} {B}x{h}x{w} mask {name}: {us:7.2f} us  {byts/1e6:6.1f} MB  {byts/us/1e6:5.2f} TB/s  frac {byts/us/1e6/8:.3f}")
