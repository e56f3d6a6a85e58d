"""In-kernel cycle stamps of the streaming 16-bit volume (a copy of the product kernel with STAMP() points): where a workgroup's
time goes — prologue, per sub-tile [wait+barrier, DMA issue, MFMA+stores], segment changes, final drain."""
import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhstream_probe.so"))
lib.run.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 6 + [ctypes.c_void_p] * 2
N, B, C = 4800, 2, 256
f1 = torch.randn(B, N, C, device="cuda").half(); f2 = torch.randn(B, N, C, device="cuda").half()
out = torch.empty(B * N * N, device="cuda")
s = torch.cuda.current_stream().cuda_stream
ts = torch.zeros(16 * 256, dtype=torch.int64, device="cuda")
def go(n):
    for _ in range(n): assert lib.run(f1.data_ptr(), f2.data_ptr(), out.data_ptr(), N, N, B, 2, 512, 54 * 1024, s, ts.data_ptr()) == 0
go(300)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); go(200); e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) * 5:.1f} us per launch (with stamps)")
a = ts.cpu().view(16, 256)
t0 = min(int(a[w, 0]) for w in range(14))
for w in range(14):
    n = int(a[w, 255]); v = (a[w, :n] - t0).tolist()
    print(f"wg {w * 37}: start {v[0]}  prologue done {v[1] - v[0]}  end {v[-1]}  (drain {v[-1] - v[-2]})  stamps {n}")
    body = v[2:-2]
    i = 0; prev = v[1]; line = []
    while i + 3 < len(body) + 1 and i + 3 <= len(body):
        w_, iss, mm = body[i] - prev, body[i + 2] - body[i + 1], body[i + 3] - body[i + 2]
        bar = body[i + 1] - body[i]
        line.append(f"[{w_}|{bar}|{iss}|{mm}]")
        prev = body[i + 3]; i += 4
    print("    [gap|wait+barrier|dma issue|mfma+stores]: " + " ".join(line))
