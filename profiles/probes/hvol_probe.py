"""A/B of the 16-bit HWC cost volume: MV_H_STREAM=1 (streaming, default) vs 0 (tile form); prints us/launch, TB/s of
algorithmic output bytes, and a checksum so the two runs can be compared bit for bit."""
import os, sys, hashlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib
ops = importlib.import_module("mac-vo_amd.ops")

def run(dt, B, H, W, C):
    g = torch.Generator(device="cuda").manual_seed(3)
    f1 = torch.randn(B, H, W, C, device="cuda", generator=g).to(dt)
    f2 = torch.randn(B, H, W, C, device="cuda", generator=g).to(dt)
    out = torch.empty(B * H * W, 1, H, W, device="cuda")
    for _ in range(300):
        ops.corr_volume(f1, f2, "hwc", out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 300
    e0.record()
    for _ in range(n):
        ops.corr_volume(f1, f2, "hwc", out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    ref = torch.einsum("bnc,bmc->bnm", f1.view(B, -1, C).float(), f2.view(B, -1, C).float()).reshape(out.shape)
    err = (out - ref).abs().max().item()
    h = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]
    byt = out.numel() * 4 + (f1.numel() + f2.numel()) * 2
    print(f"{str(dt):16s} B={B} {H}x{W} C={C}: {us:7.1f} us  {byt / us / 1e6:5.2f} TB/s  max|d|={err:.2e}  sha={h}", flush=True)

print("MV_H_STREAM =", os.environ.get("MV_H_STREAM", "(default 1)"))
for dt in (torch.float16, torch.bfloat16):
    run(dt, 2, 60, 80, 256)
run(torch.float16, 2, 90, 160, 256)
run(torch.float16, 1, 60, 80, 128)
run(torch.float16, 2, 59, 64, 256)   # N1 = 3776 = 29.5 bands: half band
