"""Host-side time per frame of the native driver, by call (is the one-lane stream host-bound?).  usage: host_breakdown.py [steps]"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from macvo_amd import ops
from macvo_amd.pipeline import Camera, FrameInputs, HotPathConfig, NativeHotPath
from tools import synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda")
cam, frames_cpu, _ = synth.make_sequence(24, 480, 640, C=256, iters=12, seed=1000, pool=2, closed_loop=True)
cache = {}
def to_dev(t):
    k = t.data_ptr()
    if k not in cache: cache[k] = t.to(dev)
    return cache[k]
frames = [FrameInputs(static=True, **{k: to_dev(v) for k, v in fr.items()}) for fr in frames_cpu]
hp = NativeHotPath(Camera(**cam), HotPathConfig(volume_precision=os.environ.get("TL_PRECISION", "f16x2")), dev)
hp.initialize(frames[0]); torch.manual_seed(0)
for _ in hp.run(frames[(1 + k) % 24] for k in range(100)): pass
acc = collections.Counter()
lib = hp._lib
class Timed:
    def __init__(self, name, fn): self.name, self.fn = name, fn
    def __call__(self, *a):
        t = time.perf_counter(); r = self.fn(*a); acc[self.name] += time.perf_counter() - t; return r
class LibProxy:
    def __getattr__(self, k):
        f = getattr(lib, k)
        return Timed(k, f) if k.startswith("mv_frame_pipe") else f
hp._lib = LibProxy()
orig_randperm = torch.randperm
def rp(*a, **k):
    t = time.perf_counter(); r = orig_randperm(*a, **k); acc["torch.randperm"] += time.perf_counter() - t; return r
torch.randperm = rp
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in hp.run(frames[(1 + k) % 24] for k in range(steps)): pass
torch.cuda.synchronize(); wall = time.perf_counter() - t0
print(f"wall {wall / steps * 1e6:.1f} us/frame = {steps / wall:.0f} frames/s")
tot = 0
for k, v in acc.most_common():
    print(f"  {k:36s} {v / steps * 1e6:7.1f} us/frame"); tot += v
print(f"  {'(python between the calls)':36s} {(wall - tot) / steps * 1e6:7.1f} us/frame")
