"""Seeded synthetic inputs shared by tests / bench / smoke.  CPU generator only."""
import torch


def flow_cov_maps(H=480, W=640, seed=2, nan_frac=0.0):
    """S-sel: sigma_uu, sigma_vv = exp(2 N(0, .5)), sigma_uv = 0  -> [1,3,H,W] float32."""
    g = torch.Generator().manual_seed(seed)
    c = torch.exp(2 * 0.5 * torch.randn(1, 2, H, W, generator=g))
    fc = torch.cat([c, torch.zeros(1, 1, H, W)], dim=1)
    if nan_frac > 0:
        m = torch.rand(1, 1, H, W, generator=g) < nan_frac
        fc[:, 0:1][m] = float("nan")
    return fc


def depth_maps(H=480, W=640, seed=3):
    """depth ~ U(1, 60) smooth-ish plane + noise; depth cov = (z^2/80)^2 * exp(N(0,.3)) -> two [1,1,H,W]."""
    g = torch.Generator().manual_seed(seed)
    ys, xs = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    z = (3 + 40 * ys + 15 * xs)[None, None] + 0.05 * torch.randn(1, 1, H, W, generator=g)
    zc = (z ** 2 / 80) ** 2 * torch.exp(0.3 * torch.randn(1, 1, H, W, generator=g)) * 1e-2
    return z.float(), zc.float()


def keypoints(n=200, H=480, W=640, seed=5, border=32):
    g = torch.Generator().manual_seed(seed)
    u = torch.randint(border, W - border, (n,), generator=g)
    v = torch.randint(border, H - border, (n,), generator=g)
    return torch.stack([u, v], dim=1)
