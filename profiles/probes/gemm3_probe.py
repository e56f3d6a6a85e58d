"""Scratch (round 2): A/B of fp32 volume GEMM variants (see gemm3_probe.hip).  usage: gemm3_probe.py [B] [modes...]"""
import ctypes as C, os, sys, statistics, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libgemm3_probe.so"))
lib.gemm3_launch.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(os.environ.get("PROBE_N", "4800"))
Cc = 256
modes = [int(a) for a in sys.argv[2:]] or [0, 1, 3, 4, 5]
torch.manual_seed(0)
f1 = torch.randn(B, Cc, N, device="cuda"); f2 = torch.randn(B, Cc, N, device="cuda"); out = torch.empty(B, N, N, device="cuda")
fl = B * 2.0 * N * N * Cc
extq = torch.zeros(1 << 16, dtype=torch.int32, device="cuda")
lib.gemm3_set_queue.argtypes = [C.c_void_p]
lib.gemm3_set_queue(extq.data_ptr())
def run(mode, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): rc = lib.gemm3_launch(f1.data_ptr(), f2.data_ptr(), out.data_ptr(), B, Cc, N, mode, None)
    e1.record(); torch.cuda.synchronize()
    assert rc == 0, (mode, rc)
    return e0.elapsed_time(e1) * 1e3 / n
for _ in range(3):
    for m in modes: run(m, 20)          # warm-up (clocks, code objects)
lib.gemm3_launch(f1.data_ptr(), f2.data_ptr(), out.data_ptr(), B, Cc, N, 0, None); torch.cuda.synchronize(); ref = out.clone()
res = {m: [] for m in modes}
for r in range(7):
    for m in modes: res[m].append(run(m, 10))
for m in modes:
    out.fill_(float("nan")); lib.gemm3_launch(f1.data_ptr(), f2.data_ptr(), out.data_ptr(), B, Cc, N, m, None); torch.cuda.synchronize()
    ok = torch.equal(out, ref)
    md, mn = statistics.median(res[m]), min(res[m])
    print(f"B={B} N={N} mode {m:3d}: median {md:8.1f} us ({fl / md / 1e6:6.1f} TF = {fl / md / 1e6 / 157.3:.3f})  min {mn:8.1f} us ({fl / mn / 1e6:6.1f} TF)  bitwise==mode0: {ok}", flush=True)
