"""Round 6 probe: does the speed of a 32-lane pipe depend on how many pipes the process created (and destroyed) before it?  (config4 leg of bench.py: 6.4 k frames/s
behind the parity legs' device-driven 1-lane pipes, 7.9 k directly.)  usage: python profiles/probes/r6_queue_history.py <n_one_lane_pipes_before> [lanes]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from macvo_amd.pipeline import Camera, FrameInputs, HotPathConfig, NativeHotPath, stack_lanes  # noqa: E402
from tools import synth  # noqa: E402

k_before = int(sys.argv[1])
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
cam, frames, _ = synth.make_sequence(6, 480, 640, C=256, iters=12, seed=1000, closed_loop=True)
ins = [FrameInputs(static=True, **{k: (None if v is None else v.to(dev)) for k, v in fr.items()}) for fr in frames]


def run(lanes_, steps, seed):
    batches = ins if lanes_ == 1 else [stack_lanes([ins[(t + l) % 6] for l in range(lanes_)]) for t in range(6)]
    hot = NativeHotPath(Camera(**cam), HotPathConfig(), dev, lanes=lanes_, generators=[seed + l for l in range(lanes_)])
    hot.initialize(batches[0])
    for _ in hot.run(batches[(1 + k) % 6] for k in range(10)):
        pass
    if os.environ.get("PROBE_COUNTS"):
        hot.enqueue_frontend(batches[1]); r = hot.finish(); r = r if isinstance(r, list) else [r]; _ = [x.n_sel for x in r]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in hot.run(batches[(1 + k) % 6] for k in range(steps)):
        pass
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del hot
    return lanes_ * steps / dt


for i in range(k_before):
    print("one-lane pipe", i, round(run(1, 200, 7 + i)), "frames/s", flush=True)
print(f"{lanes}-lane pipe after {k_before} one-lane pipes:", round(run(lanes, 60, 99)), "frames/s", flush=True)
print(f"{lanes}-lane pipe again:", round(run(lanes, 60, 99)), "frames/s", flush=True)

# ---- second part: which of bench.py's CPU-baseline / parity activities slows a later 32-lane device-driven pipe (PROBE_MODE=...)
mode = os.environ.get("PROBE_MODE", "")
if mode:
    if mode == "threads8":
        torch.set_num_threads(8)
    elif mode == "oracle":
        from oracle.pipeline import OracleHotPath
        torch.set_num_threads(8)
        ora = OracleHotPath(cam, dict(graph_type="disp"))
        ora.initialize(frames[0])
        ora.step(frames[1])
    elif mode == "extras":
        hot = NativeHotPath(Camera(**cam), HotPathConfig(), dev, generators=[1234], keep_extras=True)
        hot.initialize(ins[0])
        sink = torch.zeros(5, 7, device=dev)
        for r in hot.run((ins[1 + k] for k in range(5)), pose_sink=sink):
            hot.sync_pose()
            _ = r.kp0_uv.cpu(), r.kp0_uv[r.extras["valid"]].cpu()
        torch.cuda.synchronize()
        if os.environ.get("PROBE_CLOSE"):
            hot.close()
        del hot
    elif mode == "ops":
        from macvo_amd import ops
        vol = ops.corr_volume(ins[0].fmap1, ins[0].fmap2)
        for it in range(12):
            ops.corr_lookup(vol, ins[0].coords[it], 4)
        torch.cuda.synchronize()
    elif mode == "subprocess":
        import subprocess
        subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.is_available())"], capture_output=True)
    print(f"[{mode}] 32-lane pipe:", round(run(32, 60, 99)), "frames/s", flush=True)
