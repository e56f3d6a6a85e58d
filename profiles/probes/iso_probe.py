"""Scratch: why does the volume GEMM take 218 us inside bench.py's process and 191 us in the probe?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from macvo_amd import ops
from tests import synth
dev = torch.device("cuda")
def iso(f1, f2, tag, n=50):
    vol = ops.corr_volume(f1, f2)
    for _ in range(10): ops.corr_volume(f1, f2, out=vol)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.corr_volume(f1, f2, out=vol)
    e1.record(); torch.cuda.synchronize()
    print(f"{tag}: {e0.elapsed_time(e1) * 1e3 / n:.1f} us", flush=True)
torch.manual_seed(0)
a, b = torch.randn(2, 256, 60, 80, device=dev), torch.randn(2, 256, 60, 80, device=dev)
iso(a, b, "fresh process, randn on device")
cam, frames_cpu, _ = synth.make_sequence(24, 480, 640, C=256, iters=12, seed=1000, pool=2, closed_loop=True)
f1, f2 = frames_cpu[0]["fmap1"].to(dev), frames_cpu[0]["fmap2"].to(dev)
iso(f1, f2, "synth features")
iso(a, b, "randn again")
big = [torch.empty(1 << 30, dtype=torch.uint8, device=dev) for _ in range(8)]
iso(f1, f2, "after allocating 8 GB")
time.sleep(5)
iso(f1, f2, "after 5 s idle")
