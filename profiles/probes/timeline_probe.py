import sys, time; sys.path.insert(0, "/root/repo")
import torch
from macvo_amd import ops
from macvo_amd.pipeline import Camera, FrameInputs, HotPath, HotPathConfig
from tests import synth
dev = torch.device("cuda:0")
cam, frames_cpu, _ = synth.make_sequence(12, 480, 640, C=256, iters=12, seed=1, pool=2, closed_loop=True)
frames = [FrameInputs(static=True, **{k: v.to(dev) for k, v in fr.items()}) for fr in frames_cpu]
hot = HotPath(Camera(**cam), HotPathConfig(), dev)
torch.manual_seed(0)
hot.initialize(frames[0])
for _ in hot.run(frames[(1 + k) % 12] for k in range(24)): pass
torch.cuda.synchronize()
marks = []
def wrap(name, fn):
    def f(*a, **k):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(*a, **k); e1.record(); marks.append((name, e0, e1)); return out
    return f
ops.corr_volume = wrap("V", ops.corr_volume)
orig_lk = ops.corr_lookup
cnt = {"n": 0}
def lk(*a, **k):
    cnt["n"] += 1
    if cnt["n"] % 12 in (1, 0):
        return wrap("L%d" % (cnt["n"] % 12), orig_lk)(*a, **k)
    return orig_lk(*a, **k)
ops.corr_lookup = lk
ops.kp_select = wrap("S", ops.kp_select)
ops.kp_track = wrap("T", ops.kp_track)
ops.pgo_solve = wrap("P", ops.pgo_solve)
base = torch.cuda.Event(enable_timing=True); base.record()
t0 = time.perf_counter()
for _ in hot.run(frames[(1 + k) % 12] for k in range(12)): pass
torch.cuda.synchronize()
print("wall per frame %.1f us" % ((time.perf_counter() - t0) / 12 * 1e6))
for name, e0, e1 in marks[10:60]:
    print("%-3s %8.1f %8.1f" % (name, base.elapsed_time(e0) * 1e3, base.elapsed_time(e1) * 1e3))
