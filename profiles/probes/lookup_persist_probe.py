"""PROBE (ran once at the end of round 4: bit-identical tokens, 121-127 us vs 111 us shipped — profiles/r04_lookup_persist_probe.log): the persistent, software-pipelined batched lookup of
lookup_persist_probe.hip against mv_corr_lookup's batched kernel at BASELINE configs[4] (B = 64 pairs): bit-equality of the tokens first, then
back-to-back launch times.  Build in the container:  bash profiles/probes/lookup_persist_build.sh ;  run on the GPU box:
    python profiles/probes/lookup_persist_probe.py [B]
Expected if the idea holds: variant 0 (shipped) ~116 us at B = 64; the persistent forms below it (the kernel moves ~305 MB in that time = 2.6 TB/s:
it is latency-bound).  If the ISA still shows partial `s_waitcnt vmcnt(n)` inside the issue phase (exec-masked address regions hipcc builds around the
selects), the next step is the address arithmetic in 32-bit offsets from one SGPR base (`global_load_dword v, v_off, s[base:base+1]`) so that a masked
lane is an offset select, not a 64-bit pointer select."""
import ctypes as C
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
lib = C.CDLL(os.path.join(HERE, "liblookup_persist_probe.so"))
lib.probe_lookup_launch.argtypes = [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_void_p, C.c_void_p]
from macvo_amd import ops  # noqa: E402

dev = "cuda"
B, H, W = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 60, 80
N = H * W
g = torch.Generator(device="cpu").manual_seed(0)
vol = torch.randn(B * N, 1, H, W, device=dev)
ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
base = torch.stack([xs, ys]).float()[None].repeat(B, 1, 1, 1)
coords = [(base + (torch.rand(B, 2, H, W, generator=g) * 16 - 8)).to(dev).contiguous() for _ in range(4)]      # SURVEY §8(d) S-lookup: grid + U(-8, 8)
coords[1][:, :, ::7, ::5] = coords[1][:, :, ::7, ::5].round()                                                # integer coordinates: the margin rows
zero = torch.zeros(4, device=dev)
st = torch.cuda.current_stream().cuda_stream
refs = [ops.corr_lookup(vol, c, 4) for c in coords]
outs = [torch.empty_like(r) for r in refs]
for variant, wgs, name in ((0, 0, "shipped <4,4,32>, one group per workgroup"), (1, 2, "persistent <4,4,32>, 2 workgroups per CU"),
                           (1, 4, "persistent <4,4,32>, 4 workgroups per CU"), (1, 3, "persistent <4,4,32>, 3 workgroups per CU"),
                           (2, 1, "persistent <4,8,64>, 1 workgroup per CU"), (2, 2, "persistent <4,8,64>, 2 workgroups per CU"),
                           (10, 0, "shipped kernel, tiling <4,8,32> (256 threads)"), (11, 0, "shipped kernel, tiling <4,8,64> (512 threads)"),
                           (12, 0, "shipped kernel, tiling <4,4,16> (256 threads)"), (13, 0, "shipped kernel, tiling <4,16,64> (256 threads)"),
                           (14, 0, "shipped kernel, tiling <4,8,16> (128 threads)"), (15, 0, "shipped kernel, tiling <4,2,16> (one-frame form)"),
                           (0, 0, "shipped <4,4,32> again")):
    for i in range(4):
        outs[i].zero_()
        assert lib.probe_lookup_launch(vol.data_ptr(), coords[i].data_ptr(), outs[i].data_ptr(), B, N, H, W, variant, wgs, zero.data_ptr(), st) == 0
    torch.cuda.synchronize()
    ok = all(torch.equal(o, r) for o, r in zip(outs, refs))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        for i in range(4):
            lib.probe_lookup_launch(vol.data_ptr(), coords[i].data_ptr(), outs[i].data_ptr(), B, N, H, W, variant, wgs, zero.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
    print(f"B={B} {name:44s} {e0.elapsed_time(e1) * 1e3 / 40:8.2f} us/launch  tokens {'bit-identical' if ok else 'MISMATCH'}", flush=True)
