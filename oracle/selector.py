"""Covariance-aware keypoint selectors (TEST INFRASTRUCTURE).

Restates ``Module/KeypointSelector.py``: ``CovAwareSelector.select_point`` (:260-334),
``CovAwareSelector_NoDepth.select_point`` (:362-407) and ``MappingPointSelector.select_point``
(:87-100) with the same torch ops in the same order (max_pool2d NMS, (nan)median, strict ``<``
thresholds, ``nonzero`` row-major order, CPU ``torch.randperm`` from the global generator,
``roll`` to (u, v)).  PINNED against the real reference modules by ``tests/golden/make_golden.py``.

Every function returns ``(pixels_uv int64 [n,2], candidates_vu int64 [m,2], aux dict)`` where
``candidates_vu`` is ``torch.nonzero(point_mask)[:, 2:]`` before the random permutation.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _border(like: torch.Tensor, mask_width: int) -> torch.Tensor:
    border = torch.zeros_like(like, dtype=torch.bool)
    border[..., mask_width:-mask_width, mask_width:-mask_width] = True
    return border


def _finish(point_mask: torch.Tensor, numPoint: int):
    selected = torch.nonzero(point_mask, as_tuple=False)
    perm = torch.randperm(selected.size(0))[:numPoint]  # global CPU generator, as the reference
    pixels = selected[perm][..., 2:].roll(shifts=1, dims=1)
    return pixels, selected[..., 2:]


def cov_aware_selector_nodepth(flow_cov: torch.Tensor, numPoint: int, kernel_size: int = 7, mask_width: int = 32,
                               max_match_cov: float = 100.0, match_mask: torch.Tensor | None = None):
    """``CovAwareSelector_NoDepth.select_point`` (``KeypointSelector.py:362-407``).

    flow_cov: ``[1,3,H,W]`` float32 (sigma_uu, sigma_vv, sigma_uv).
    """
    quality_map = (flow_cov[:, 0] + flow_cov[:, 1] - 2 * flow_cov[:, 2]).unsqueeze(1)
    flow_cov_map = quality_map
    erode = -F.max_pool2d(-quality_map, kernel_size=kernel_size, stride=1, padding=kernel_size // 2)
    quality_nms = torch.logical_and(quality_map == erode, ~quality_map.isnan())
    border_mask = _border(quality_nms, mask_width)
    median = flow_cov_map[quality_nms].median().item()
    flow_cov_thresh = min(max_match_cov, median * 1.5)
    flow_cov_mask = flow_cov_map < flow_cov_thresh
    point_mask = torch.logical_and(quality_nms, border_mask)
    point_mask = torch.logical_and(point_mask, flow_cov_mask)
    if match_mask is not None:
        point_mask = torch.logical_and(point_mask, match_mask)
    pixels, cand = _finish(point_mask, numPoint)
    return pixels, cand, {"median": median, "thresh": flow_cov_thresh, "n_nms": int(quality_nms.sum())}


def cov_aware_selector(depth0: torch.Tensor, depth0_cov: torch.Tensor, depth1: torch.Tensor, depth1_cov: torch.Tensor,
                       flow_cov: torch.Tensor | None, numPoint: int, max_depth: float, kernel_size: int = 7,
                       mask_width: int = 32, max_depth_cov: float = 250.0, max_match_cov: float = 100.0,
                       depth0_mask: torch.Tensor | None = None, match_mask: torch.Tensor | None = None):
    """``CovAwareSelector.select_point`` (``KeypointSelector.py:260-334``).  ``max_depth`` is the resolved
    value (``"auto"`` -> fx * baseline, ``:263``).  All maps ``[1,1,H,W]`` float32, flow_cov ``[1,3,H,W]``."""
    quality_map = depth0_cov + depth1_cov
    flow_cov_map = None
    if flow_cov is not None:
        flow_cov_map = (flow_cov[:, 0] + flow_cov[:, 1] - 2 * flow_cov[:, 2]).unsqueeze(1)
        quality_map = quality_map * flow_cov_map
    erode = -F.max_pool2d(-quality_map, kernel_size=kernel_size, stride=1, padding=kernel_size // 2)
    quality_nms = torch.logical_and(quality_map == erode, ~quality_map.isnan())
    border_mask = _border(quality_nms, mask_width)
    depth_mask = (depth0 < max_depth) & (depth1 < max_depth)
    med_d = depth0_cov[quality_nms].nanmedian().item()
    depth0_cov_thresh = min(max_depth_cov, med_d * 1.5)
    depth0_cov_mask = depth0_cov < depth0_cov_thresh
    aux = {"median_depth_cov": med_d, "depth_cov_thresh": depth0_cov_thresh, "n_nms": int(quality_nms.sum())}
    flow_cov_mask = None
    if flow_cov_map is not None:
        med_f = flow_cov_map[quality_nms].nanmedian().item()
        flow_cov_thresh = min(max_match_cov, med_f * 1.5)
        flow_cov_mask = flow_cov_map < flow_cov_thresh
        aux.update(median_flow_cov=med_f, flow_cov_thresh=flow_cov_thresh)
    point_mask = torch.logical_and(quality_nms, border_mask)
    point_mask = torch.logical_and(point_mask, depth_mask)
    point_mask = torch.logical_and(point_mask, depth0_cov_mask)
    if flow_cov_mask is not None:
        point_mask = torch.logical_and(point_mask, flow_cov_mask)
    if depth0_mask is not None:
        point_mask = torch.logical_and(point_mask, depth0_mask)
    if match_mask is not None:
        point_mask = torch.logical_and(point_mask, match_mask)
    pixels, cand = _finish(point_mask, numPoint)
    return pixels, cand, aux


def mapping_point_selector(depth0: torch.Tensor, depth0_cov: torch.Tensor, numPoint: int, max_depth: float = 5.0,
                           max_depth_cov: float = 0.005, mask_width: int = 32):
    """``MappingPointSelector.select_point`` (``KeypointSelector.py:87-100``)."""
    depth_mask = depth0 < max_depth
    depth_cov_mask = depth0_cov < max_depth_cov
    border_mask = _border(depth_mask, mask_width)
    candidates = depth_mask & depth_cov_mask & border_mask
    pixels, cand = _finish(candidates, numPoint)
    return pixels, cand, {}
