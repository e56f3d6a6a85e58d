"""In-kernel s_memtime stamps of corr_volume_split_stream (probe library, MV_SPLIT_DBG=16): per item and wave
[t0 before barrier H0, t1 after it, t2 end of half 0, t3 before barrier H1, t4 after, t5 end of half 1]."""
import ctypes as C, os, sys
os.environ["MV_SPLIT_DBG"] = os.environ.get("MV_SPLIT_DBG", "16")
here = os.path.dirname(os.path.abspath(__file__))
os.environ["MACVO_HIP_LIB"] = os.path.join(here, "libmacvo_hip_split_probe.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
import numpy as np, torch
from macvo_amd import ops, _lib
B, C_, H, W = 2, 256, 60, 80
N = H * W
g = torch.Generator().manual_seed(0)
f1 = torch.randn(B, C_, H, W, generator=g).cuda(); f2 = torch.randn(B, C_, H, W, generator=g).cuda()
pk = ops.volume_pack(f1, f2)
out = torch.empty((B * N, 1, H, W), device="cuda")
for _ in range(200):
    ops.corr_volume_packed(pk[0], pk[1], B, C_, N, N, out=out)
torch.cuda.synchronize()
lib = C.CDLL(os.environ["MACVO_HIP_LIB"])
n = 256 * 4 * 64 * 8
buf = np.zeros(n, dtype=np.int64)
lib.mv_split_probe_stamps.argtypes = [C.c_void_p, C.c_size_t]
rc = lib.mv_split_probe_stamps(buf.ctypes.data, n)
print("rc", rc)
st = buf.reshape(256, 4, 64, 8)
for wg in (0, 1, 37, 255):
    for wave in (0, 3):
        s = st[wg, wave]
        k = int((s[:, 5] != 0).sum())
        t00 = s[0, 0]
        print(f"wg {wg} wave {wave}: {k} items, total {(s[k-1,5]-t00)} ticks")
        for i in range(k):
            r = s[i]
            print(f"   item {i:2d} (it {r[6]:5d}) start {r[0]-t00:7d} | barrier0 {r[1]-r[0]:5d} mfma0 {r[2]-r[1]:5d} | barrier1 {r[4]-r[3]:5d} mfma1 {r[5]-r[4]:5d} | gap to next {(s[i+1,0]-r[5]) if i+1<k else 0:5d}")
# aggregate over all waves
tot = []; bar = []; mf = []; gap = []
for wg in range(256):
    for wave in range(4):
        s = st[wg, wave]; k = int((s[:, 5] != 0).sum())
        if k < 2: continue
        tot.append(s[k-1,5]-s[0,0]); bar.append(((s[:k,1]-s[:k,0])+(s[:k,4]-s[:k,3])).sum()); mf.append(((s[:k,2]-s[:k,1])+(s[:k,5]-s[:k,4])).sum()); gap.append((s[1:k,0]-s[:k-1,5]).sum() + (s[:k,3]-s[:k,2]).sum())
print(f"mean over waves: total {np.mean(tot):.0f} ticks, barriers {np.mean(bar):.0f}, mfma phases {np.mean(mf):.0f}, gaps {np.mean(gap):.0f}; s_memtime runs at 100 MHz")
