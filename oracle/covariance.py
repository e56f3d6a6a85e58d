"""2D -> 3D covariance propagation (TEST INFRASTRUCTURE).

Restates ``Module/Covariance/Project2to3.py`` (``MatchCovariance.estimate`` :124-181,
``Covariance_2to3_full`` :377-423, ``create_3x3_matrix``/``create_2x2_matrix`` :426-443) and
``Utility/Math.py:43-63`` (``gaussain_full_kernels``) with the same torch ops in the same order,
including the reference's quirks (in-place clamp of the caller's ``flow_cov``; Gaussian kernel
applied transposed relative to the depth patch; result assembled on CPU and cast to float64).
PINNED against the real reference modules by ``tests/golden/make_golden.py``.
"""
from __future__ import annotations

import torch


def create_2x2_matrix(m, n_sample: int, device) -> torch.Tensor:
    mat = torch.empty((n_sample, 2, 2), device=device)
    for i in range(2):
        for j in range(2):
            mat[..., i, j] = m[i][j]
    return mat


def create_3x3_matrix(m, n_sample: int) -> torch.Tensor:
    mat = torch.empty((n_sample, 3, 3))
    for i in range(3):
        for j in range(3):
            mat[..., i, j] = m[i][j]
    return mat


def gaussain_full_kernels(cov_2x2: torch.Tensor, kernel_size: int) -> torch.Tensor:
    """``Utility/Math.py:43-63`` (eager semantics; the reference wraps it in ``torch.compile``)."""
    N = cov_2x2.size(0)
    det_cov = cov_2x2.det()
    inv_cov = cov_2x2.pinverse().float()
    x = torch.linspace(-(kernel_size - 1) / 2.0, (kernel_size - 1) / 2.0, kernel_size, device=cov_2x2.device)
    y = torch.linspace(-(kernel_size - 1) / 2.0, (kernel_size - 1) / 2.0, kernel_size, device=cov_2x2.device)
    indices = torch.stack(torch.meshgrid(x, y, indexing="ij"), dim=-1).unsqueeze(0).repeat(N, 1, 1, 1)
    z = torch.einsum("bxyi,bij,bxyj->bxy", indices, -0.5 * inv_cov, indices).exp()
    kernel = z / (2 * torch.pi * torch.sqrt(det_cov)).view(N, 1, 1)
    kernel_s = kernel.sum(dim=[-1, -2], keepdim=True)
    return kernel / kernel_s


def Covariance_2to3_full(sigma_uu, sigma_uv, sigma_vv, sigma_dd, u, v, d, fx, fy, cx, cy) -> torch.Tensor:
    """``Project2to3.py:377-423`` — 3x3 in NED order (z, x, y)."""
    sigma_xx = (((u - cx).square() * sigma_dd) + (d.square() * sigma_uu) + (sigma_uu * sigma_dd)) / (fx ** 2)
    sigma_yy = (((v - cy).square() * sigma_dd) + (d.square() * sigma_vv) + (sigma_vv * sigma_dd)) / (fy ** 2)
    sigma_zz = sigma_dd
    sigma_xy = (((u - cx) * (v - cy) * sigma_dd) + (d.square() + sigma_dd) * sigma_uv) / (fx * fy)
    sigma_xz = (sigma_dd * (u - cx)) / fx
    sigma_yz = (sigma_dd * (v - cy)) / fy
    return create_3x3_matrix(
        [[sigma_zz, sigma_xz, sigma_yz], [sigma_xz, sigma_xx, sigma_xy], [sigma_yz, sigma_xy, sigma_yy]],
        n_sample=u.size(0),
    )


def match_covariance(kp: torch.Tensor, depth_map: torch.Tensor, depth_cov: torch.Tensor | None,
                     flow_cov: torch.Tensor | None, fx: float, fy: float, cx: float, cy: float,
                     kernel_size: int = 31, match_cov_default: float = 0.25, min_flow_cov: float = 0.25,
                     min_depth_cov: float = 0.05, return_aux: bool = False):
    """``MatchCovariance.estimate`` (``Project2to3.py:124-181``).

    kp ``[N,2]`` (u, v) any dtype; depth_map ``[1,1,H,W]`` float32; depth_cov ``[N]`` or None;
    flow_cov ``[N,3]`` or None — **clamped in place** like the reference (:131).
    Returns ``[N,3,3]`` float64 on CPU.
    """
    n_sample = kp.size(0)
    hlf = kernel_size // 2
    kp_long = kp.clone().long()
    has_flow_cov = flow_cov is not None
    if has_flow_cov:
        flow_cov[..., :2].clamp_(min=min_flow_cov ** 2)
    else:
        flow_cov = torch.ones((n_sample, 3), dtype=torch.float) * match_cov_default
        flow_cov[..., 2] = 0.0
    var_u, var_v, var_uv = flow_cov[..., 0], flow_cov[..., 1], flow_cov[..., 2]
    kp_u, kp_v = kp[..., 0], kp[..., 1]

    u_idx = torch.arange(-hlf, hlf + 1, dtype=torch.long)
    v_idx = torch.arange(-hlf, hlf + 1, dtype=torch.long)
    uu, vv = torch.meshgrid(u_idx, v_idx, indexing="ij")
    all_u = kp_long[:, 0].unsqueeze(-1) + uu.reshape(1, -1)
    all_v = kp_long[:, 1].unsqueeze(-1) + vv.reshape(1, -1)

    cov_m = create_2x2_matrix([[var_u, var_uv], [var_uv, var_v]], n_sample, device=depth_map.device)
    local_filters = gaussain_full_kernels(cov_m, kernel_size)
    patches = depth_map[..., all_v, all_u].view(n_sample, kernel_size, kernel_size)
    patches = patches.permute(0, 2, 1)

    wavg = (local_filters * patches).sum(dim=[1, 2])
    if has_flow_cov or (depth_cov is None):
        wvar = torch.sum(local_filters * (patches - wavg.unsqueeze(1).unsqueeze(1)).square(), dim=[1, 2])
    else:
        wvar = depth_cov
    wvar = wvar.clamp(min=min_depth_cov)
    cov = Covariance_2to3_full(var_u, var_uv, var_v, wvar, kp_u, kp_v, wavg, fx, fy, cx, cy).double()
    if return_aux:
        return cov, {"wavg": wavg, "wvar": wvar, "filters": local_filters}
    return cov


def rotate_covariance(R: torch.Tensor, cov: torch.Tensor) -> torch.Tensor:
    """``Odometry/MACVO.py:273-281``: cov_Tw = R_prev @ cov_Tc @ R_prev^T in float64 (bmm order kept)."""
    n = cov.size(0)
    Rn = R.repeat((n, 1, 1)).to(torch.float64)
    return torch.bmm(torch.bmm(Rn, cov), Rn.transpose(1, 2))
