import ctypes as C, os, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmfma_peak.so"))
lib.mfma_spin_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
out = torch.zeros(16, device="cuda")
for blocks, nacc in ((256, 4), (512, 4), (1024, 4), (1024, 2), (2048, 4)):
    for iters in (2000, 8000):
        lib.mfma_spin_launch(out.data_ptr(), blocks, 100, nacc, None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.mfma_spin_launch(out.data_ptr(), blocks, iters, nacc, None); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        flops = blocks * 4 * iters * 8 * nacc * 4096.0
        print(f"blocks {blocks:5d} waves/SIMD {blocks*4/1024:.0f} nacc {nacc} iters {iters}: {ms*1e3:9.1f} us  {flops/ms/1e9:7.1f} TFLOP/s")
