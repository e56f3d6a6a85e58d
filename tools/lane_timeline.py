"""Unprofiled GPU timeline of the native frame driver for a given number of lanes (HIP events the driver records itself):
period, GEMM duration, idle on the GEMM stream, GEMM end -> last lookup, -> selector done.  usage: lane_timeline.py [lanes] [steps]"""
import os, sys, statistics as st, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macvo_amd.pipeline import Camera, FrameInputs, HotPathConfig, NativeHotPath, stack_lanes
from tools import synth
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else (200 if lanes < 8 else 40)
pool = 24
dev = torch.device("cuda")
cam, frames_cpu, _ = synth.make_sequence(pool, 480, 640, C=256, iters=12, seed=1000, pool=2, closed_loop=True)
cache = {}
def to_dev(t):
    k = t.data_ptr()
    if k not in cache: cache[k] = t.to(dev)
    return cache[k]
frames = [FrameInputs(static=True, **{k: to_dev(v) for k, v in fr.items()}) for fr in frames_cpu]
batches = frames if lanes == 1 else [stack_lanes([frames[(t + l) % pool] for l in range(lanes)]) for t in range(pool)]
gens = None if lanes == 1 else [torch.Generator().manual_seed(l) for l in range(lanes)]
hp = NativeHotPath(Camera(**cam), HotPathConfig(volume_precision=os.environ.get("TL_PRECISION", "bf16x3")), dev, lanes=lanes, generators=gens)
hp.initialize(batches[0]); torch.manual_seed(0)
warm = max(10, steps // 4)
for _ in hp.run(batches[(1 + k) % pool] for k in range(warm)): pass
import time
sink = torch.zeros((steps, 7) if lanes == 1 else (steps, lanes, 7), device=dev) if os.environ.get("TL_SINK") else None
hp.time_volume(steps)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in hp.run((batches[(1 + warm + k) % pool] for k in range(steps)), pose_sink=sink): pass
torch.cuda.synchronize(); wall = time.perf_counter() - t0
print(f"wall: {wall / steps * 1e6:.1f} us/step = {lanes * steps / wall:.0f} frames/s over {steps} steps (incl. fill / drain)")
tl = hp.timeline_ms()
lo, hi = len(tl) // 4, len(tl) - 2
per = [(tl[i + 1][0] - tl[i][0]) * 1e3 for i in range(lo, hi)]
gap = [(tl[i + 1][0] - tl[i][1]) * 1e3 for i in range(lo, hi)]
gem = [(tl[i][1] - tl[i][0]) * 1e3 for i in range(lo, hi)]
lk = [(tl[i][2] - tl[i][1]) * 1e3 for i in range(lo, hi)]
sel = [(tl[i][3] - tl[i][2]) * 1e3 for i in range(lo, hi)]
print(f"lanes {lanes}: period {st.median(per):.1f} us ({lanes * 1e6 / st.median(per):.0f} frames/s) | GEMM {st.median(gem):.1f} | idle on the GEMM stream "
      f"before the next GEMM {st.median(gap):.1f} | GEMM end -> last lookup done {st.median(lk):.1f} | -> selector done {st.median(sel):.1f}")
for i in range(lo, lo + 3):
    b = tl[i][0]
    print("  step", i, [round((x - b) * 1e3, 1) for x in tl[i]], "next GEMM start", round((tl[i + 1][0] - b) * 1e3, 1))
