"""A/B of the fp32 CHW cost volume: MV_VOL_STREAM=1 (streaming) vs default (mixed-tile); us/launch, TF/s, checksum."""
import os, sys, hashlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from macvo_amd import ops

def run(B, H, W, C):
    g = torch.Generator(device="cuda").manual_seed(3)
    f1 = torch.randn(B, C, H, W, device="cuda", generator=g)
    f2 = torch.randn(B, C, H, W, device="cuda", generator=g)
    out = torch.empty(B * H * W, 1, H, W, device="cuda")
    for _ in range(150):
        ops.corr_volume(f1, f2, "chw", out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 150
    e0.record()
    for _ in range(n):
        ops.corr_volume(f1, f2, "chw", out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    h = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]
    fl = 2.0 * B * (H * W) ** 2 * C
    print(f"B={B} {H}x{W} C={C}: {us:7.1f} us  {fl / us / 1e6:6.1f} TF/s ({fl / us / 1e6 / 157.3:.3f})  sha={h}", flush=True)

print("MV_VOL_STREAM =", os.environ.get("MV_VOL_STREAM", "(default 0)"), " REGIONS =", os.environ.get("MV_VOL_STREAM_REGIONS", "auto"))
run(2, 60, 80, 256)
if not os.environ.get("QUICK"):
    run(2, 59, 64, 256)
    run(1, 64, 64, 256)
    run(2, 90, 160, 256)
