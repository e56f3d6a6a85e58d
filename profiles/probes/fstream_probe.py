"""Decomposition of the fp32 streaming volume (copy of the product kernel with DBG bits): 1 = 1/16 of the stores, 2 = no DMA in
the loop, 4 = no LDS fragment reads, 8 = no barriers, 16 = A fragments loaded once only."""
import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfstream_probe.so"))
lib.run.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
N, B, C = 4800, 2, 256
f1 = torch.randn(B, C, N, device="cuda"); f2 = torch.randn(B, C, N, device="cuda")
out = torch.empty(B * N * N, device="cuda")
s = torch.cuda.current_stream().cuda_stream
R = 3
def t(dbg, n=100):
    for _ in range(100): lib.run(dbg, f1.data_ptr(), f2.data_ptr(), out.data_ptr(), N, B, R, 512, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): assert lib.run(dbg, f1.data_ptr(), f2.data_ptr(), out.data_ptr(), N, B, R, 512, s) == 0
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
names = {0: "full", 1: "1/16 of the stores", 2: "no DMA", 3: "no stores, no DMA", 4: "no LDS reads", 7: "MFMA + barriers (+A loads)", 15: "MFMA + A loads", 31: "MFMA only", 16: "full, A loaded once"}
for dbg in names:
    print(f"dbg {dbg:2d} {names[dbg]:24s}: {t(dbg):6.1f} us", flush=True)
