import ctypes as C, os, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmfma16_mix.so"))
lib.mix_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
out = torch.zeros(256 * 256 * 64 + 64, device="cuda")
cyc = torch.zeros(1, dtype=torch.int64, device="cuda")
data = torch.randn(1024 * 8).to(torch.bfloat16).cuda()
names = {1: "B from LDS", 2: "AGPR stores", 4: "2 s_nop per pair", 8: "4 accumulators"}
for flags in (0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 15):
    iters = 200
    for _ in range(2):
        assert lib.mix_launch(data.data_ptr(), out.data_ptr(), iters, flags, cyc.data_ptr(), None) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lib.mix_launch(data.data_ptr(), out.data_ptr(), iters, flags, cyc.data_ptr(), None); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    n = iters * 192
    print(f"flags {flags:2d} [{', '.join(v for k, v in names.items() if flags & k) or 'bare MFMA stream'}]: {us:8.1f} us -> {cyc.item() / n:.1f} cycles/MFMA, {256 * 4 * n * 32768 / us / 1e6:.0f} TFLOP/s", flush=True)
