"""Fill / drain of a 20-step timed region (the driver's --steps 20): timeline of every step relative to the region's start."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from macvo_amd.pipeline import Camera, FrameInputs, HotPathConfig, NativeHotPath
from tests import synth
steps, pool = 20, 24
dev = torch.device("cuda")
cam, frames_cpu, _ = synth.make_sequence(pool, 480, 640, C=256, iters=12, seed=1000, pool=2, closed_loop=True)
frames = [FrameInputs(static=True, **{k: v.to(dev) for k, v in fr.items()}) for fr in frames_cpu]
hp = NativeHotPath(Camera(**cam), HotPathConfig(), dev)
hp.initialize(frames[0]); torch.manual_seed(0)
for _ in hp.run(frames[(1 + k) % pool] for k in range(300)): pass
for rep in range(3):
    sink = torch.zeros((steps, 7), device=dev)
    hp.time_volume(steps)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    host = []
    for _ in hp.run((frames[(1 + k) % pool] for k in range(steps)), pose_sink=sink):
        host.append((time.perf_counter() - t0) * 1e6)
    t_loop = (time.perf_counter() - t0) * 1e6
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e6
    tl = hp.timeline_ms()
    b = tl[0][0]
    print(f"rep {rep}: wall {wall:.0f} us = {wall / steps:.1f} us/step; host loop returned at {t_loop:.0f} us")
    print("   GEMM start:", [round((x[0] - b) * 1e3) for x in tl])
    print("   GEMM end  :", [round((x[1] - b) * 1e3) for x in tl])
    print("   lookups dn:", [round((x[2] - b) * 1e3) for x in tl])
    print("   select dn :", [round((x[3] - b) * 1e3) for x in tl])
    print("   host yield:", [round(x) for x in host])
