"""Observation filters of ``Module/OutlierFilter.py`` (TEST INFRASTRUCTURE — CPU restatement, pinned bit-exactly against the
real classes by tests/golden/filters.npz)."""
from __future__ import annotations

import torch


def covariance_sanity(cov1: torch.Tensor, cov2: torch.Tensor) -> torch.Tensor:
    """``CovarianceSanityFilter.filter`` (OutlierFilter.py:91-100): drop rows whose 3x3 covariances hold a NaN or an inf."""
    bad = cov1.isnan().any(dim=[-1, -2]) | cov1.isinf().any(dim=[-1, -2]) | cov2.isnan().any(dim=[-1, -2]) | cov2.isinf().any(dim=[-1, -2])
    return ~bad


def simple_depth(d1: torch.Tensor, d2: torch.Tensor, min_depth: float, max_depth: float) -> torch.Tensor:
    """``SimpleDepthFilter.filter`` (OutlierFilter.py:112-116): keep min <= d <= max for both observations (strict rejects)."""
    return ~((d1 < min_depth) | (d1 > max_depth) | (d2 < min_depth) | (d2 > max_depth)).squeeze(-1)


def likely_front_of_cam(d1: torch.Tensor, c1: torch.Tensor, d2: torch.Tensor, c2: torch.Tensor) -> torch.Tensor:
    """``LikelyFrontOfCamFilter.filter`` (OutlierFilter.py:131-141): both depths stay positive at -2 sigma; if ANY
    ``pixel1_d_cov`` is the -1 placeholder ("no covariance estimate") every row passes."""
    if (c1 == -1).any():
        return torch.ones((d1.shape[0],), dtype=torch.bool)
    return (((d1 - (c1.sqrt() * 2)) > 0.) & ((d2 - (c2.sqrt() * 2)) > 0.)).squeeze(-1)
