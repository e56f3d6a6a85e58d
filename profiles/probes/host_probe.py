import sys, time; sys.path.insert(0, "/root/repo")
import torch
from macvo_amd.pipeline import Camera, FrameInputs, HotPath, HotPathConfig
from tests import synth
dev = torch.device("cuda:0")
cam, frames_cpu, _ = synth.make_sequence(6, 480, 640, C=256, iters=12, seed=1, pool=2, closed_loop=True)
frames = [FrameInputs(static=True, **{k: v.to(dev) for k, v in fr.items()}) for fr in frames_cpu]
import os
hot = HotPath(Camera(**cam), HotPathConfig(use_graphs=bool(int(os.environ.get('G','0')))), dev)
torch.manual_seed(0)
hot.initialize(frames[0])
for _ in hot.run(frames[(1 + k) % 6] for k in range(12)): pass
torch.cuda.synchronize()
# host cost of each half (GPU kept busy/async)
N = 100
te = tf = 0.0
pend = hot.enqueue_frontend(frames[1])
for i in range(N):
    t0 = time.perf_counter(); nxt = hot.enqueue_frontend(frames[(i + 2) % 6]); t1 = time.perf_counter()
    hot.finish(pend); t2 = time.perf_counter()
    te += t1 - t0; tf += t2 - t1; pend = nxt
torch.cuda.synchronize()
print(f"host per frame: enqueue_frontend {te/N*1e6:.0f} us, finish (incl. waiting for the count) {tf/N*1e6:.0f} us")
