import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from macvo_amd import ops
def run(B, H, W, C=256):
    g = torch.Generator(device="cuda").manual_seed(3)
    f1 = torch.randn(B, C, H, W, device="cuda", generator=g)
    f2 = torch.randn(B, C, H, W, device="cuda", generator=g)
    N = H * W
    out = torch.full((B * N, 1, H, W), float("nan"), device="cuda")
    ops.corr_volume(f1, f2, "chw", out=out)
    torch.cuda.synchronize()
    ref = torch.einsum("bcn,bcm->bnm", f1.view(B, C, N), f2.view(B, C, N))
    d = (out.view(B, N, N) - ref).abs()
    bad = (d > 1e-2) | torch.isnan(d)
    nb = int(bad.sum())
    print(f"B={B} N={N}: bad={nb} of {bad.numel()}  max|d|={float(torch.nan_to_num(d, nan=1e9).max()):.3g}")
    if nb:
        idx = bad.nonzero()
        for b in range(B):
            m = idx[idx[:, 0] == b]
            if len(m):
                r, c = m[:, 1], m[:, 2]
                print(f"   pair {b}: {len(m)} bad; rows [{int(r.min())},{int(r.max())}] cols [{int(c.min())},{int(c.max())}]  row blocks(128) {sorted(set((r // 128).tolist()))[:12]} col blocks {sorted(set((c // 128).tolist()))[:12]}")
                bb = bad[b].view(-1)
                blk = bad[b][: (N // 128) * 128, : (N // 128) * 128].view(N // 128, 128, N // 128, 128).any(3).any(1)
                print("   bad 128x128 blocks:", int(blk.sum()), "of", blk.numel(), " first:", blk.nonzero()[:10].tolist())
run(1, 64, 64)
run(2, 64, 64)
run(1, 60, 80)
run(2, 60, 80)
