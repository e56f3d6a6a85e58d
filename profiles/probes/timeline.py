"""Scratch: unprofiled GPU timeline of the native driver (HIP events recorded by the driver itself)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from macvo_amd.pipeline import Camera, FrameInputs, HotPathConfig, NativeHotPath
from tests import synth
dev = torch.device("cuda")
cam, frames_cpu, _ = synth.make_sequence(24, 480, 640, C=256, iters=12, seed=1000, pool=2, closed_loop=True)
frames = [FrameInputs(static=True, **{k: v.to(dev) for k, v in fr.items()}) for fr in frames_cpu]
hp = NativeHotPath(Camera(**cam), HotPathConfig(), dev)
hp.initialize(frames[0]); torch.manual_seed(0)
for _ in hp.run(frames[(1 + k) % 24] for k in range(40)): pass
hp.time_volume(200)
for _ in hp.run(frames[(41 + k) % 24] for k in range(200)): pass
tl = hp.timeline_ms()
import statistics as st
per = [(tl[i + 1][0] - tl[i][0]) * 1e3 for i in range(50, 190)]
gap = [(tl[i + 1][0] - tl[i][1]) * 1e3 for i in range(50, 190)]
gem = [(tl[i][1] - tl[i][0]) * 1e3 for i in range(50, 190)]
lk = [(tl[i][2] - tl[i][1]) * 1e3 for i in range(50, 190)]
sel = [(tl[i][3] - tl[i][2]) * 1e3 for i in range(50, 190)]
print(f"period {st.median(per):.1f} us | GEMM {st.median(gem):.1f} | idle on the GEMM stream before the next GEMM {st.median(gap):.1f} | GEMM end -> last lookup done {st.median(lk):.1f} | -> selector done {st.median(sel):.1f}")
for i in range(100, 104):
    b = tl[i][0]
    print("frame", i, [round((x - b) * 1e3, 1) for x in tl[i]], "next GEMM start", round((tl[i + 1][0] - b) * 1e3, 1))
