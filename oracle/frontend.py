"""Frontend epilogues, gathers and projection helpers (TEST INFRASTRUCTURE).

Restates the pure-torch pieces around the learned frontend:
``Module/Frontend/StereoDepth.py:270-282``, ``Module/Frontend/Frontend.py:103-118,183-200``,
``Module/Frontend/Matching.py:28-40``, ``Utility/Point.py:5-21``,
``Odometry/MACVO.py:197-240``.
"""
from __future__ import annotations

import torch


def disparity_to_depth(disp: torch.Tensor, bl: float, fx: float) -> torch.Tensor:
    """``StereoDepth.py:270-272``: z = (bl*fx) * 1/d  (python-float product, fp32 reciprocal, fp32 mul)."""
    return (bl * fx) * disp.reciprocal()


def disparity_to_depth_cov(disp: torch.Tensor, disp_cov: torch.Tensor, bl: float, fx: float) -> torch.Tensor:
    """``StereoDepth.py:275-282``: sigma_z^2 = (bl*fx)^2 * ((sigma_d^2 / d^2) / d^2), in that op order."""
    disparity_2 = disp.square()
    error_rate_2 = disp_cov * disparity_2.reciprocal()
    return ((bl * fx) ** 2) * (error_rate_2 / disparity_2)


def inference_2_depth(flow: torch.Tensor, cov: torch.Tensor, bl: float, fx: float, enforce_positive_disparity: bool = False):
    """``Frontend.py:183-194``: sample-0 of the batched inference -> (depth, depth_cov, disparity, disparity_cov, bad_mask)."""
    disparity, disparity_cov = flow[:, :1].abs(), cov[:, :1]
    depth = disparity_to_depth(disparity, bl, fx)
    depth_cov = disparity_to_depth_cov(disparity, disparity_cov, bl, fx)
    mask = (flow[:, :1] <= 0) if enforce_positive_disparity else None
    return depth, depth_cov, disparity, disparity_cov, mask


def from_partial_cov(cov2: torch.Tensor) -> torch.Tensor:
    """``Matching.py:28-40``: pad (sigma_uu, sigma_vv) with a zero sigma_uv channel -> [B,3,H,W]."""
    B, C, H, W = cov2.shape
    assert C == 2
    return torch.cat([cov2, torch.zeros((B, 1, H, W)).to(cov2)], dim=1)


def retrieve_pixels(pixel_uv: torch.Tensor, scalar_map: torch.Tensor) -> torch.Tensor:
    """``Frontend.py:103-118``: ``map[0, :, v.long(), u.long()]`` -> [C, N] (batch 0 only, truncation toward 0)."""
    return scalar_map[0, ..., pixel_uv[..., 1].long(), pixel_uv[..., 0].long()]


def filterPointsInRange(pts: torch.Tensor, u_range: tuple[int, int], v_range: tuple[int, int]) -> torch.Tensor:
    """``Utility/Point.py:5-13``: strict ``min < coord < max`` on both axes."""
    u_min, u_max = u_range
    v_min, v_max = v_range
    u_sel = torch.logical_and(pts[..., 0] < u_max, pts[..., 0] > u_min)
    v_sel = torch.logical_and(pts[..., 1] < v_max, pts[..., 1] > v_min)
    return torch.logical_and(u_sel, v_sel)


def pixel2point_NED(pixels: torch.Tensor, depths: torch.Tensor, K: torch.Tensor) -> torch.Tensor:
    """``Utility/Point.py:15-17``: ``pp.pixel2point(pixels, depths, K).roll(1, -1)``.  PyPose 0.6.8
    ``pixel2point`` computes EDN ``(((u-cx)*d)/fx, ((v-cy)*d)/fy, d)`` in that op order; the roll gives
    NED ``(z_cam, x_cam, y_cam)``."""
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    x = ((pixels[..., 0] - cx) * depths) / fx
    y = ((pixels[..., 1] - cy) * depths) / fy
    return torch.stack([depths, x, y], dim=-1)


def point2pixel_NED(points: torch.Tensor, K: torch.Tensor) -> torch.Tensor:
    """``Utility/Point.py:19-21``: ``pp.point2pixel(points.roll(-1, -1), K)``.  PyPose 0.6.8 computes
    ``homo2cart(p_EDN @ K^T)``: u = (fx*Y + cx*X)/X, v = (fy*Z + cy*X)/X for NED points (X = depth,
    skew K[0,1] = 0 asserted by the reference, ``Graphs.py:162``); ``homo2cart`` guards the divisor
    with ``sign(X) * max(|X|, tiny)``."""
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    X, Y, Z = points[..., 0], points[..., 1], points[..., 2]
    tiny = torch.finfo(points.dtype).tiny
    den = X.abs().clamp(min=tiny)
    den = torch.where(X >= 0, den, -den)
    u = (fx * Y + cx * X) / den
    v = (fy * Z + cy * X) / den
    return torch.stack([u, v], dim=-1)


def track_keypoints(kp0_uv: torch.Tensor, flow: torch.Tensor, cov3: torch.Tensor,
                    depth0: dict, depth1: dict, edge: int, H: int, W: int, match_cov_default: float):
    """``Odometry/MACVO.py:197-232``: kp1 = kp0 + flow[kp0]; strict border filter on kp1; the eight
    scalar-map gathers (kp0 at integer pixels, kp1 truncated toward zero); default sigma for kp0;
    match covariance read at the SOURCE pixel kp0.

    depth0 / depth1: dicts with keys depth, disparity, disparity_uncertainty, cov (each [1,1,H,W]).
    Returns a dict of per-keypoint tensors on the inputs' device.
    """
    kp1_uv = kp0_uv + retrieve_pixels(kp0_uv, flow).T
    inb = filterPointsInRange(kp1_uv, (edge, W - edge), (edge, H - edge))
    kp0_uv, kp1_uv = kp0_uv[inb], kp1_uv[inb]
    out = {"kp0_uv": kp0_uv, "kp1_uv": kp1_uv, "inbound_mask": inb}
    for tag, kp, d in (("kp0", kp0_uv, depth0), ("kp1", kp1_uv, depth1)):
        out[f"{tag}_d"] = retrieve_pixels(kp, d["depth"]).squeeze(0)
        out[f"{tag}_disparity"] = retrieve_pixels(kp, d["disparity"])
        out[f"{tag}_sigma_disparity"] = retrieve_pixels(kp, d["disparity_uncertainty"])
        out[f"{tag}_sigma_dd"] = retrieve_pixels(kp, d["cov"]).squeeze(0)
    n = kp0_uv.size(0)
    s0 = torch.ones((n, 3), device=kp0_uv.device) * match_cov_default
    s0[..., 2] = 0.0
    out["kp0_sigma_uv"] = s0
    out["kp1_sigma_uv"] = retrieve_pixels(kp0_uv, cov3).T
    return out


def upsample_flow(flow: torch.Tensor, mask: torch.Tensor, scale: float = 8) -> torch.Tensor:
    """FlowFormer/RAFT convex 8x upsampling (call sites ``covhead.py:124-126,133-135``; in-tree twin
    ``Module/Network/PWCNet/pwc_cov/gru.py:40-52``): softmax over 9 taps of the 3x3 neighbourhood of
    ``8*flow``; ``[N,2,H,W],[N,576,H,W] -> [N,2,8H,8W]``.  ``scale``: RAFT/FlowFormer multiply the coarse flow by 8;
    the in-tree PWC twin multiplies by its kernel size (3) — with ``scale=3`` this function IS that twin, which is how it
    is pinned (tests/golden/upsample.npz)."""
    N, C, H, W = flow.shape
    mask = mask.view(N, 1, 9, 8, 8, H, W)
    mask = torch.softmax(mask, dim=2)
    up = torch.nn.functional.unfold(scale * flow, [3, 3], padding=1)
    up = up.view(N, C, 9, 1, 1, H, W)
    up = torch.sum(mask * up, dim=2)
    up = up.permute(0, 1, 4, 2, 5, 3)
    return up.reshape(N, C, 8 * H, 8 * W)


def upsample_and_epilogue(flow8: torch.Tensor, mask_flow: torch.Tensor, cov8: torch.Tensor, mask_cov: torch.Tensor):
    """Last decoder iteration's tail (``covhead.py:119-140`` + ``flownet.py:44``): convex-upsample the 1/8-res flow with
    ``0.25 * up_mask`` already applied by the caller for the flow branch (:121) and the cov branch's mask as produced
    by ``CovUpdateBlock`` (0.25 applied inside, :41); returns ``(flow_up, exp(2 * cov_up))``."""
    flow_up = upsample_flow(flow8, mask_flow)
    cov_up = upsample_flow(cov8, mask_cov)
    return flow_up, torch.exp(cov_up * 2)
