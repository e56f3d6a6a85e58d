import sys; sys.path.insert(0, "/root/repo")
import torch
from macvo_amd import ops
from oracle import corr
dev = torch.device("cuda:0")
for shape in [(2, 64, 8, 12), (1, 256, 60, 80)]:
    B, C, H, W = shape
    g = torch.Generator().manual_seed(5)
    f1, f2 = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
    ref64 = corr.corr_volume(f1, f2, torch.float64)
    a1, a2 = f1.permute(0, 2, 3, 1).contiguous().to(dev), f2.permute(0, 2, 3, 1).contiguous().to(dev)
    out = ops.corr_volume(a1, a2, layout="hwc", precision="split3").cpu().double()
    exact = ops.corr_volume(a1, a2, layout="hwc").cpu().double()
    ref32 = corr.corr_volume(f1, f2, torch.float32).double()
    scale = (f1.double().abs().reshape(B, C, -1).permute(0, 2, 1).unsqueeze(2) * f2.double().abs().reshape(B, C, -1).permute(0, 2, 1).unsqueeze(1)).sum(-1).reshape(ref64.shape)
    for name, o in (("split3", out), ("exact", exact), ("cpu_f32", ref32)):
        e = (o - ref64).abs()
        print(shape, name, "max abs", e.max().item(), "max rel-to-sum|a||b|", (e / scale).max().item(), "rms", e.pow(2).mean().sqrt().item())
