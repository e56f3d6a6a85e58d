"""Probe (round 6): does the history of destroyed pipes change a new pipe's speed?  One-lane pipe (three normal-priority streams + one high-priority) timed
(a) first in the process, (b) after a destroyed 3-lane pipe (two normal + two high-priority streams), (c) after another one-lane pipe.
Run on the GPU box: python profiles/probes/r6_pipe_order.py [first|after_batched|after_one_lane]"""
import os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tools import synth
from macvo_amd.pipeline import Camera, FrameInputs, HotPathConfig, NativeHotPath, stack_lanes

dev = torch.device("cuda:0")
cam, frames_cpu, _ = synth.make_sequence(12, 480, 640, C=256, iters=12, seed=5, pool=6, closed_loop=True)
frames = [FrameInputs(static=True, **{k: v.to(dev) for k, v in fr.items()}) for fr in frames_cpu]


def run(lanes, steps, seed):
    batches = frames if lanes == 1 else [stack_lanes([frames[(t + l) % len(frames)] for l in range(lanes)]) for t in range(len(frames))]
    hot = NativeHotPath(Camera(**cam), HotPathConfig(), dev, lanes=lanes, generators=[seed + l for l in range(lanes)])
    hot.initialize(batches[0])
    for _ in hot.run(batches[(1 + k) % len(batches)] for k in range(100)):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in hot.run(batches[(5 + k) % len(batches)] for k in range(steps)):
        pass
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    hot.close()
    del hot
    return lanes * steps / dt


mode = sys.argv[1] if len(sys.argv) > 1 else "first"
if mode == "after_batched":
    print("3-lane pipe first:", round(run(3, 60, 1)), "frames/s")
elif mode == "after_one_lane":
    print("one-lane pipe first:", round(run(1, 300, 1)), "frames/s")
print(mode, "-> one-lane pipe:", round(run(1, 300, 2)), "frames/s", "GPU_MAX_HW_QUEUES=" + os.environ.get("GPU_MAX_HW_QUEUES", "default"))
