import sys, os; sys.path.insert(0, "/root/repo")
import torch
from macvo_amd import ops
from oracle import corr
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, C, h, w = 2, 256, 60, 80
f1, f2 = torch.randn(B, C, h, w, generator=g).to(dev), torch.randn(B, C, h, w, generator=g).to(dev)
vol_a = ops.corr_volume(f1, f2)
coords = (corr.coords_grid(B, h, w) + 3.0).to(dev)
tok1 = torch.empty((B, 81, h, w), device=dev); tok2 = torch.empty_like(tok1)
big = torch.empty(64 * 1024 * 1024, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
def run(fa, fb):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    e[0].record()
    with torch.cuda.stream(sa):
        sa.wait_event(e[0]); fa(); e[1].record(sa)
    with torch.cuda.stream(sb):
        sb.wait_event(e[0]); fb(); e[2].record(sb)
    torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) * 1e3, e[0].elapsed_time(e[2]) * 1e3
lk1 = lambda: [ops.corr_lookup(vol_a, coords, 4, out=tok1) for _ in range(12)]
lk2 = lambda: [ops.corr_lookup(vol_a, coords, 4, out=tok2) for _ in range(12)]
fill = lambda: [big.fill_(1.0) for _ in range(4)]
nop = lambda: None
for name, fa, fb in (("lookups alone", lk1, nop), ("lookups || lookups", lk1, lk2), ("fill alone", fill, nop), ("fill || lookups", fill, lk2)):
    for _ in range(2): run(fa, fb)
    a, b = run(fa, fb)
    print(f"{name:22s}: stream A done {a:7.1f} us, stream B done {b:7.1f} us   (GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')})")
