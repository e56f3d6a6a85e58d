import ctypes as C, os, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmfma16_probe.so"))
lib.spin_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
out = torch.zeros(16, device="cuda")
cyc = torch.zeros(1, dtype=torch.int64, device="cuda")
for name, data in (("random", torch.randn(1024 * 8).to(torch.bfloat16).cuda()), ("zeros", torch.zeros(1024 * 8, dtype=torch.bfloat16).cuda())):
    for agpr in (0, 1, 2):
        for nacc in ((1, 2, 4, 8) if agpr < 2 else (2, 4)):
            na = 8 if agpr < 2 else 48
            iters = 4000 if agpr < 2 else 700
            for _ in range(2):
                lib.spin_launch(data.data_ptr(), out.data_ptr(), 256, iters, nacc, agpr, cyc.data_ptr(), None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); lib.spin_launch(data.data_ptr(), out.data_ptr(), 256, iters, nacc, agpr, cyc.data_ptr(), None); e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3
            n = iters * na * nacc
            print(f"{name:6s} A in {'VGPR' if agpr == 0 else 'AGPR'} x{na:2d} chains {nacc}: {us:8.1f} us, {n} MFMAs/wave -> {us * 1e3 / n:.2f} ns/MFMA, "
                  f"{cyc.item() / n:.1f} s_memtime ticks/MFMA, {256 * 4 * n * 32768 / us / 1e6:.0f} TFLOP/s", flush=True)
