import ctypes as C, os, sys
here = os.path.dirname(os.path.abspath(__file__))
os.environ["MACVO_HIP_LIB"] = os.path.join(here, "libmacvo_hip_pgostamps.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
import numpy as np, torch
from macvo_amd import ops
from oracle import pgo
from tests.test_gpu_backend import _to_batch
prob, _ = pgo.make_synthetic_problem(n=200, seed=6)
b = _to_batch([prob], torch.device("cuda"))
for _ in range(20):
    pose, info = ops.pgo_solve(b, "disp")
torch.cuda.synchronize()
print("steps", info[0, 1].item(), "rejects", info[0, 2].item())
lib = C.CDLL(os.environ["MACVO_HIP_LIB"])
buf = np.zeros(128, dtype=np.int64)
lib.mv_pgo_probe_stamps.argtypes = [C.c_void_p]
print("rc", lib.mv_pgo_probe_stamps(buf.ctypes.data))
st = buf.reshape(16, 8)
names = ["accumulate (skipped behind an accepted fused trial)", "reduction (28 / 55 values)", "clamp+damp+cholesky", "se3 update", "trial: build at the trial pose + quality term", "its 29-value reduction", "TR + accept (+ reject rounds)"]
for k in range(int(info[0, 1].item())):
    r = st[k]
    print(f"step {k}: " + " | ".join(f"{names[i]} {r[i+1]-r[i]}" for i in range(7)) + f" | total {r[7]-r[0]}" + (f" | to next {st[k+1][0]-r[7]}" if k + 1 < 16 and st[k+1][0] else ""))
rn = ["replay", "cholesky", "se3", "four 64-point loss passes", "quality + TR + hand-over", "walk"]
for k in (14, 15):
    r = st[k]
    if r[0]:
        print(f"reject round {k - 13} (wave 3): " + " | ".join(f"{rn[i]} {r[i+1]-r[i]}" for i in range(6)) + f" | total {r[6]-r[0]}")
