import ctypes as C, os, sys, statistics, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgemm_probe.so"))
lib.gemm_probe_launch.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]
B, N, Cc = 2, 4800, 256
f1 = torch.randn(B, Cc, N, device="cuda"); f2 = torch.randn(B, Cc, N, device="cuda"); out = torch.empty(B, N, N, device="cuda")
fl = B * 2.0 * N * N * Cc
rgs = [int(a) for a in sys.argv[1:]] or [5, 8, 10, 13, 19, 38]
def run(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): lib.gemm_probe_launch(f1.data_ptr(), f2.data_ptr(), out.data_ptr(), B, Cc, N, 16, None)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
res = {r: [] for r in rgs}
for r in rgs:
    lib.gemm_probe_set_rg(r); run(10)
for rnd in range(6):
    for r in rgs:
        lib.gemm_probe_set_rg(r); torch.cuda.synchronize(); res[r].append(run(10))
for r in rgs:
    md = statistics.median(res[r]); print(f"RG {r:3d}: median {md:7.1f} us ({fl / md / 1e6:6.1f} TF) min {min(res[r]):7.1f}")
