"""In-kernel cycle stamps of the fp32 streaming volume: per item [pre | wait+barrier h0 | half 0 | wait+barrier h1 | half 1 + stores]."""
import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfstamp_probe.so"))
lib.run.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
N, B, C = 4800, 2, 256
f1 = torch.randn(B, C, N, device="cuda"); f2 = torch.randn(B, C, N, device="cuda")
out = torch.empty(B * N * N, device="cuda")
s = torch.cuda.current_stream().cuda_stream
ts = torch.zeros(16 * 256, dtype=torch.int64, device="cuda")
def go(n):
    for _ in range(n): assert lib.run(f1.data_ptr(), f2.data_ptr(), out.data_ptr(), N, B, 3, 512, s, ts.data_ptr()) == 0
go(100)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); go(100); e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) * 10:.1f} us per launch (with stamps)")
a = ts.cpu().view(16, 256)
for w in range(14):
    n = int(a[w, 255]); v = (a[w, :n] - a[w, 0]).tolist()
    body = v[1:-1]
    line = []
    prev = 0
    for i in range(0, len(body) - 5, 6):
        s0, a0, b0, a1, b1, e = body[i:i + 6]
        line.append(f"[{s0 - prev}|{b0 - a0}|{a1 - b0}|{b1 - a1}|{e - b1}]")
        prev = e
    print(f"wg {w * 37}: total {v[-1]}  items {len(line)}\n    " + " ".join(line))
