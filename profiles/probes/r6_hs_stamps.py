"""Round 6 probe: s_memtime stamps inside corr_volume_h_stream (library built with -DMV_HS_STAMPS): per item and wave [enter, barrier passed, DMA issued, MFMAs + stores issued].
usage: MACVO_HIP_LIB=<probe lib> python profiles/probes/r6_hs_stamps.py [out16|fp32] [B]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from macvo_amd import _lib as L, ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "out16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lib = C.CDLL(os.environ["MACVO_HIP_LIB"])
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
a1 = torch.randn(B, 60, 80, 256, generator=g).half().to(dev)
a2 = torch.randn(B, 60, 80, 256, generator=g).half().to(dev)
buf = torch.zeros(32 * 4 * 48 * 4, dtype=torch.int64, device=dev)
run = (lambda: ops.corr_volume_out16(a1, a2)) if mode == "out16" else (lambda: ops.corr_volume(a1, a2, "hwc"))
for _ in range(20):
    run()
torch.cuda.synchronize()
assert lib.mv_hs_probe_stamps(C.c_void_p(buf.data_ptr())) == 0
run()
torch.cuda.synchronize()
lib.mv_hs_probe_stamps(None)
st = buf.cpu().view(32, 4, 48, 4)
tick = 100e6            # s_memtime: 100 MHz constant clock on gfx9 (10 ns)
import statistics as S
rows = []
for wg in range(32):
    for w in range(4):
        x = st[wg, w]
        n = int((x[:, 3] != 0).sum())
        for i in range(1, n - 1):       # steady items (skip the first and the last)
            rows.append(((x[i, 1] - x[i, 0]).item(), (x[i, 2] - x[i, 1]).item(), (x[i, 3] - x[i, 2]).item(), (x[i + 1, 0] - x[i, 3]).item(), (x[i + 1, 0] - x[i, 0]).item()))
print(f"{mode} B={B}: {len(rows)} steady items sampled; items per wave ~{n}")
names = ["wait+barrier", "DMA issue", "MFMA+stores", "gap to next item", "item total"]
for k, nm in enumerate(names):
    v = [r[k] for r in rows]
    print(f"  {nm:18s} median {S.median(v) / tick * 1e6:7.2f} us   mean {S.mean(v) / tick * 1e6:7.2f} us   max {max(v) / tick * 1e6:7.2f} us")
w0 = st[0, 0]
print("  wg 0 wave 0 first items (us since its first stamp): " + " | ".join("%.2f %.2f %.2f %.2f" % tuple(((w0[i, j] - w0[0, 0]).item() / tick * 1e6) for j in range(4)) for i in range(min(6, n))))
