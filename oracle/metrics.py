"""Trajectory metrics (TEST INFRASTRUCTURE): relative translation / rotation error between two pose tracks.

Restates what ``Evaluation/MetricsSeq.py:9-16`` (``evaluateRTE``) and ``:26-41`` (``evaluateROE``) ask of
``evo.main_rpe.rpe(pose_relation=translation_part | rotation_angle_deg, delta=1, delta_unit=frames)`` — evo 1.x is an
un-vendored dependency (``requirements.txt``), so its published definition is restated:

    E_i = (Q_i^-1 Q_{i+1})^-1 (P_i^-1 P_{i+1}),   RTE_i = ||trans(E_i)||,   ROE_i = angle(rot(E_i))

with Q the reference track and P the estimate.  evo first aligns the estimate to the reference (origin / Umeyama,
``align=True``); a rigid alignment ``P_i -> S P_i`` cancels inside ``P_i^-1 P_{i+1}``, so without scale correction
(``correct_scale=False`` for MAC-VO, ``NEED_ALIGN_SCALE`` :7) the relative errors do not depend on it and it is omitted.
Poses are ``[T, 7]`` ``(tx ty tz qx qy qz qw)``.
"""
from __future__ import annotations

import torch

from . import se3


def relative_errors(ref: torch.Tensor, est: torch.Tensor, delta: int = 1):
    """-> (rte [T-delta] metres, roe [T-delta] radians)."""
    ref, est = ref.double(), est.double()
    assert ref.shape == est.shape and ref.shape[-1] == 7 and ref.shape[0] > delta
    dq = se3.se3_mul(se3.se3_inv(ref[:-delta]), ref[delta:])
    dp = se3.se3_mul(se3.se3_inv(est[:-delta]), est[delta:])
    e = se3.se3_mul(se3.se3_inv(dq), dp)
    return e[..., :3].norm(dim=-1), se3.so3_log(e[..., 3:]).norm(dim=-1)


def rte(ref: torch.Tensor, est: torch.Tensor, delta: int = 1) -> dict:
    """evo's RPE statistics of the translation part (``result.stats``: mean / rmse / max), in metres."""
    t, r = relative_errors(ref, est, delta)
    return {"mean": float(t.mean()), "rmse": float(t.square().mean().sqrt()), "max": float(t.max()),
            "roe_mean_rad": float(r.mean()), "roe_max_rad": float(r.max()), "pairs": int(t.numel())}
