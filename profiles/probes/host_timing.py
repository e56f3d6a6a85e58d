"""Scratch: where does the host spend a native-driver frame?  (perf_counter around the three per-frame calls)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from macvo_amd import ops
from macvo_amd.pipeline import Camera, FrameInputs, HotPathConfig, NativeHotPath
from tests import synth
dev = torch.device("cuda")
cam, frames_cpu, _ = synth.make_sequence(24, 480, 640, C=256, iters=12, seed=1000, pool=2, closed_loop=True)
frames = [FrameInputs(static=True, **{k: v.to(dev) for k, v in fr.items()}) for fr in frames_cpu]
hp = NativeHotPath(Camera(**cam), HotPathConfig(), dev)
hp.initialize(frames[0])
torch.manual_seed(0)
L, lib, C = ops.L, hp._lib if hp._pipe else None, ops.C
def loop(n, rec):
    t_idx = 1
    hp.enqueue_frontend(frames[t_idx % 24]); t_idx += 1
    for i in range(n):
        t0 = time.perf_counter()
        hp.enqueue_frontend(frames[t_idx % 24]); t_idx += 1
        t1 = time.perf_counter()
        L.check(hp._lib.mv_frame_pipe_wait_candidates(hp._pipe, C.byref(hp._ncand)), "w")
        t2 = time.perf_counter()
        perm = torch.randperm(hp._ncand.value)[:200]
        t3 = time.perf_counter()
        L.check(hp._lib.mv_frame_pipe_finish(hp._pipe, perm.data_ptr(), perm.numel(), None), "f")
        hp._n_fin += 1
        t4 = time.perf_counter()
        if rec is not None: rec.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
    # drain the one still pending
    hp.finish()
    hp.synchronize()
loop(30, None)
rec = []
t0 = time.perf_counter(); loop(300, rec); el = time.perf_counter() - t0
import statistics as st
cols = list(zip(*rec))
print(f"frame {el / 301 * 1e6:.1f} us | enqueue {st.median(cols[0])*1e6:.1f}  wait_candidates {st.median(cols[1])*1e6:.1f}  randperm {st.median(cols[2])*1e6:.1f}  finish {st.median(cols[3])*1e6:.1f} (medians, us)")
