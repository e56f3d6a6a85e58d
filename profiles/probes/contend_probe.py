"""Why do the window lookups run 3x slower beside the volume GEMM?  12 back-to-back lookups on stream B while stream A runs
(a) nothing, (b) the volume GEMM into another buffer, (c) an MFMA-only spin kernel (no memory traffic, 2 waves / SIMD), (d) a
store-only kernel writing 184 MB in the GEMM's pattern.  Reports the lookups' time per call and stream A's time."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from macvo_amd import ops
from oracle import corr
here = os.path.dirname(os.path.abspath(__file__))
mf = C.CDLL(os.path.join(here, "libmfma_peak.so")); mf.mfma_spin_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
st = C.CDLL(os.path.join(here, "libstore_probe.so")); st.run.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, Cc, h, w = 2, 256, 60, 80
f1, f2 = torch.randn(B, Cc, h, w, generator=g).to(dev), torch.randn(B, Cc, h, w, generator=g).to(dev)
vol_a = ops.corr_volume(f1, f2); vol_b = torch.empty_like(vol_a)
coords = [(corr.coords_grid(B, h, w) + 3.0 + 0.37 * i).to(dev) for i in range(12)]
tok = torch.empty((B, 81, h, w), device=dev)
spin_out = torch.zeros(16, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def heavy(kind, n):
    for _ in range(n):
        if kind == "gemm": ops.corr_volume(f1, f2, out=vol_b)
        elif kind == "mfma": mf.mfma_spin_launch(spin_out.data_ptr(), 512, 110, 4, sa.cuda_stream)     # ~200 us of MFMA only, 2 waves / SIMD, 4 chains
        elif kind == "mfma1": mf.mfma_spin_launch(spin_out.data_ptr(), 256, 220, 4, sa.cuda_stream)    # 1 wave / SIMD
        elif kind == "mfma2c": mf.mfma_spin_launch(spin_out.data_ptr(), 512, 110, 2, sa.cuda_stream)   # 2 waves / SIMD, 2 chains each
        elif kind == "store": st.run(0, 1, vol_b.data_ptr(), 4800, 4800, 2, 512, sa.cuda_stream)
def measure(kind):
    n_heavy = {"none": 0, "gemm": 14, "mfma": 14, "mfma1": 14, "mfma2c": 14, "store": 60}[kind]
    res = []
    for rep in range(4):
        e0 = torch.cuda.Event(enable_timing=True); ea = torch.cuda.Event(enable_timing=True)
        eb0 = torch.cuda.Event(enable_timing=True); eb1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        with torch.cuda.stream(sa):
            sa.wait_event(e0); heavy(kind, n_heavy); ea.record(sa)
        with torch.cuda.stream(sb):
            sb.wait_event(e0)
            for _ in range(3):                       # let stream A get going first
                for c in coords: ops.corr_lookup(vol_a, c, 4, out=tok)
            eb0.record(sb)
            for _ in range(4):
                for c in coords: ops.corr_lookup(vol_a, c, 4, out=tok)
            eb1.record(sb)
        torch.cuda.synchronize()
        res.append((eb0.elapsed_time(eb1) * 1e3 / 48, e0.elapsed_time(ea) * 1e3 / max(n_heavy, 1), e0.elapsed_time(eb1) * 1e3, e0.elapsed_time(ea) * 1e3))
    r = res[-1]
    print(f"beside {kind:6s}: lookup {r[0]:6.1f} us/call   stream A {r[1]:7.1f} us/launch   (lookups finished at {r[2]:.0f} us, stream A at {r[3]:.0f} us)")
for kind in ("none", "gemm", "mfma", "mfma1", "mfma2c", "store"):
    measure(kind)
