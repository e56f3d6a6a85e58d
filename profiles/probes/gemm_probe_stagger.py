import ctypes as C, os, sys, statistics, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgemm_probe.so"))
lib.gemm_probe_launch.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]
B, N, Cc = 2, 4800, 256
f1 = torch.randn(B, Cc, N, device="cuda"); f2 = torch.randn(B, Cc, N, device="cuda"); out = torch.empty(B, N, N, device="cuda")
fl = B * 2.0 * N * N * Cc
def run(mode, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): lib.gemm_probe_launch(f1.data_ptr(), f2.data_ptr(), out.data_ptr(), B, Cc, N, mode, None)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
cfgs = [(16, 0)] + [(18, s) for s in (1, 2, 3, 4, 6)]
for m, s in cfgs:
    lib.gemm_probe_set_stagger(s); run(m, 10)
res = {c: [] for c in cfgs}
for rnd in range(6):
    for c in cfgs:
        lib.gemm_probe_set_stagger(c[1]); torch.cuda.synchronize(); res[c].append(run(c[0], 10))
for c in cfgs:
    md = statistics.median(res[c]); print(f"mode {c[0]} stagger {c[1]} x 3.4 us per slot: median {md:7.1f} us ({fl / md / 1e6:6.1f} TF) min {min(res[c]):7.1f}")
