"""Scratch: where do the 8.7 us of a one-frame window lookup go?  (run on the GPU box)"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblookup_probe.so"))
lib.probe_launch.argtypes = [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p]
dev = "cuda"
B, H, W = (int(sys.argv[1]) if len(sys.argv) > 1 else 2), 60, 80
N = H * W
vol = torch.randn(B, N, H, W, device=dev)
g = torch.Generator(device="cpu").manual_seed(0)
ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
base = torch.stack([xs, ys]).float()[None].repeat(B, 1, 1, 1)
coords = [(base + torch.randn(B, 2, H, W, generator=g) * 3).to(dev).contiguous() for _ in range(12)]
outs = [torch.empty(B, 81, H, W, device=dev) for _ in range(12)]
st = torch.cuda.current_stream().cuda_stream
names = {0: "empty 300x1024", 1: "QPW2 QPB32 (shipped small)", 2: "QPW2 QPB16", 3: "QPW1 QPB16", 4: "QPW4 QPB32", 5: "QPW1 QPB8",
         6: "QPW2 QPB8", 7: "QPW4 QPB16", 8: "QPW8 QPB32 (shipped large)", 9: "QPW4 QPB64", 10: "QPW1 QPB4", 11: "QPW2 QPB4"}
from macvo_amd import ops
ref = ops.corr_lookup(vol.view(B * N, 1, H, W), coords[5], 4)
for v in range(12):
    for _ in range(3):
        for i in range(12):
            assert lib.probe_launch(vol.data_ptr(), coords[i].data_ptr(), outs[i].data_ptr(), B, N, H, W, v, st) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        for i in range(12):
            lib.probe_launch(vol.data_ptr(), coords[i].data_ptr(), outs[i].data_ptr(), B, N, H, W, v, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 240
    chk = "" if v == 0 else ("match" if torch.equal(outs[5], ref.view_as(outs[5])) else "MISMATCH")
    print(f"variant {v:2d} {names[v]:28s} {us:7.2f} us/launch (back-to-back) {chk}")
