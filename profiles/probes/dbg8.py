import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from macvo_amd import ops
for (B, H, W) in ((2, 16, 24), (1, 60, 80)):
    g = torch.Generator().manual_seed(0)
    f1 = torch.randn(B, 256, H, W, generator=g).cuda(); f2 = torch.randn(B, 256, H, W, generator=g).cuda()
    N = H * W
    ref = ops.corr_volume(f1, f2).view(B, N, N)
    out = ops.corr_volume(f1, f2, precision="f16x2").view(B, N, N)
    err = (out - ref).abs()
    bad = err > 1e-3
    print(B, H, W, "max err", err.max().item(), "bad frac", bad.float().mean().item())
    if bad.any():
        idx = bad.nonzero()
        print(" first bad", idx[:5].tolist(), " rows%128 hist", torch.bincount((idx[:, 1] % 128) // 32, minlength=4).tolist(),
              " cols%64 hist", torch.bincount((idx[:, 2] % 64) // 32, minlength=2).tolist(), " pairs", torch.bincount(idx[:, 0], minlength=B).tolist())
        b, i, j = idx[0].tolist()
        print(" sample out", out[b, i, j].item(), "ref", ref[b, i, j].item(), "ratio", (out[b, i, j] / ref[b, i, j]).item())
        # are bad entries in specific sub-tiles?
        print(" bad sub-tile cols (j//64)", torch.unique(idx[:, 2] // 64).tolist()[:20], " bands", torch.unique(idx[:, 1] // 128).tolist()[:20])
