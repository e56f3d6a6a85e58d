import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libstore_probe.so"))
lib.run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
N, B = 4800, 2
out = torch.empty(B * N * N, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def t(mode, ncb, grid, n=200):
    for _ in range(300): lib.run(mode, ncb, out.data_ptr(), N, N, B, grid, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): assert lib.run(mode, ncb, out.data_ptr(), N, N, B, grid, s) == 0
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"mode {mode} ncb {ncb} grid {grid:5d}: {us:6.1f} us  {out.numel() * 4 / us / 1e6:.2f} TB/s", flush=True)
for _ in range(300): out.fill_(1.0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): out.fill_(1.0)
e1.record(); torch.cuda.synchronize()
print(f"torch fill: {e0.elapsed_time(e1) * 10:.1f} us")
lib.run_rd.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
big = torch.zeros(1 << 29, device="cuda")          # 2 GB
def tr(rd, lines, grid=512, n=200):
    for _ in range(300): lib.run_rd(rd, out.data_ptr(), big.data_ptr(), lines - 1, N, N, B, grid, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): assert lib.run_rd(rd, out.data_ptr(), big.data_ptr(), lines - 1, N, N, B, grid, s) == 0
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"reads {rd * 4:3d} KB/item ({rd * 4 * 5700 / 1e3:6.1f} MB/launch) from {lines * 128 / 2**20:7.1f} MB buffer: {us:6.1f} us", flush=True)
for lines in (1 << 12, 1 << 18, 1 << 24):          # 512 KB (L2), 32 MB (MALL), 2 GB (HBM)
    for rd in (0, 1, 2, 4, 8, 16):
        tr(rd, lines)
