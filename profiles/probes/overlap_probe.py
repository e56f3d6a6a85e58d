import sys, os; sys.path.insert(0, "/root/repo")
import torch
from macvo_amd import ops
from oracle import corr
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, C, h, w = 2, 256, 60, 80
f1, f2 = torch.randn(B, C, h, w, generator=g).to(dev), torch.randn(B, C, h, w, generator=g).to(dev)
vol_a = ops.corr_volume(f1, f2); vol_b = torch.empty_like(vol_a)
coords = (corr.coords_grid(B, h, w) + 3.0).to(dev)
tok = torch.empty((B, 81, h, w), device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
torch.cuda.synchronize()
def run(n_lk):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    e[0].record()
    with torch.cuda.stream(sa):
        sa.wait_event(e[0]); e[1].record(sa); ops.corr_volume(f1, f2, out=vol_b); e[2].record(sa)
    with torch.cuda.stream(sb):
        sb.wait_event(e[0]); e[3].record(sb)
        for _ in range(n_lk): ops.corr_lookup(vol_a, coords, 4, out=tok)
        e[4].record(sb)
    torch.cuda.synchronize()
    return [e[0].elapsed_time(x) * 1e3 for x in (e[1], e[2], e[3], e[4])]
for _ in range(3): run(12)
for n in (1, 12):
    r = run(n)
    print(f"lookups={n:2d}: volume [{r[0]:7.1f}, {r[1]:7.1f}] us   lookups [{r[2]:7.1f}, {r[3]:7.1f}] us  (MV_LOOKUP_SMALL={os.environ.get('MV_LOOKUP_SMALL')})")
