"""SE(3)/SO(3) arithmetic restated from PyPose 0.6.8 (TEST INFRASTRUCTURE).

``pypose`` is an un-vendored dependency of the reference (``requirements.txt:1``,
CI pin ``static_analysis_requirements.txt:1``) and is not installable here, so the
pieces the hot path uses are restated from PyPose's published algorithm
(``pypose/lietensor/operation.py``): "parity unpinned" for this file.

Storage convention (reference ``Module/Map/VisualMap.py:26``): ``[tx, ty, tz, qx, qy, qz, qw]``.
Tangent convention (``Module/Optimization/TwoFramePGO/Graphs.py:193-194,224-225``):
``[rho(3), phi(3)]`` translation first, left perturbation ``T <- Exp(delta) * T``.
All functions are batched over leading dims and dtype-generic (the solver uses float64,
``Module/Optimization/TwoFramePGO/Optimizer.py:85``).
"""
from __future__ import annotations

import torch


def vec2skew(v: torch.Tensor) -> torch.Tensor:
    """pp.vec2skew — [..., 3] -> [..., 3, 3] with skew(v) @ w = v x w."""
    z = torch.zeros_like(v[..., 0])
    return torch.stack(
        [
            torch.stack([z, -v[..., 2], v[..., 1]], dim=-1),
            torch.stack([v[..., 2], z, -v[..., 0]], dim=-1),
            torch.stack([-v[..., 1], v[..., 0], z], dim=-1),
        ],
        dim=-2,
    )


def so3_exp(phi: torch.Tensor) -> torch.Tensor:
    """PyPose ``so3_Exp``: axis-angle [..., 3] -> quaternion [..., 4] (x, y, z, w).

    Small-angle branch switches at ``theta > eps(dtype)`` exactly as PyPose does.
    """
    eps = torch.finfo(phi.dtype).eps
    theta = phi.norm(dim=-1, keepdim=True)
    theta2 = theta * theta
    theta4 = theta2 * theta2
    half = 0.5 * theta
    safe = torch.where(theta > eps, theta, torch.ones_like(theta))
    imag = torch.where(theta > eps, half.sin() / safe, 0.5 - theta2 / 48.0 + theta4 / 3840.0)
    real = torch.where(theta > eps, half.cos(), 1.0 - theta2 / 8.0 + theta4 / 384.0)
    return torch.cat([phi * imag, real], dim=-1)


def so3_Jl(phi: torch.Tensor) -> torch.Tensor:
    """PyPose ``so3_Jl`` left Jacobian: I + c1*K + c2*K@K."""
    eps = torch.finfo(phi.dtype).eps
    K = vec2skew(phi)
    theta = phi.norm(dim=-1, keepdim=True).unsqueeze(-1)
    theta2 = theta * theta
    safe = torch.where(theta > eps, theta, torch.ones_like(theta))
    safe2 = safe * safe
    c1 = torch.where(theta > eps, (1.0 - safe.cos()) / safe2, 0.5 - theta2 / 24.0)
    c2 = torch.where(theta > eps, (safe - safe.sin()) / (safe * safe2), 1.0 / 6.0 - theta2 / 120.0)
    I = torch.eye(3, dtype=phi.dtype, device=phi.device).expand(K.shape)
    return I + c1 * K + c2 * (K @ K)


def quat_mul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """PyPose ``SO3_Mul`` on (x, y, z, w) quaternions."""
    av, aw = a[..., :3], a[..., 3:]
    bv, bw = b[..., :3], b[..., 3:]
    v = aw * bv + bw * av + torch.linalg.cross(av, bv)
    w = aw * bw - (av * bv).sum(dim=-1, keepdim=True)
    return torch.cat([v, w], dim=-1)


def quat_act(q: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    """PyPose ``SO3_Act``: rotate points ``p`` [..., 3] by ``q``."""
    qv, qw = q[..., :3], q[..., 3:]
    uv = torch.linalg.cross(qv.expand(p.shape), p)
    uv = uv + uv
    return p + qw * uv + torch.linalg.cross(qv.expand(p.shape), uv)


def quat_inv(q: torch.Tensor) -> torch.Tensor:
    return torch.cat([-q[..., :3], q[..., 3:]], dim=-1)


def quat_to_matrix(q: torch.Tensor) -> torch.Tensor:
    """``LieTensor.rotation().matrix()`` — unit quaternion (x, y, z, w) -> [..., 3, 3]."""
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = torch.stack(
        [
            torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], dim=-1),
            torch.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], dim=-1),
            torch.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=-1),
        ],
        dim=-2,
    )
    return R


def se3_exp(xi: torch.Tensor) -> torch.Tensor:
    """PyPose ``se3_Exp``: [rho, phi] [..., 6] -> SE3 [..., 7] = [Jl(phi) @ rho, so3_Exp(phi)]."""
    rho, phi = xi[..., :3], xi[..., 3:6]
    t = (so3_Jl(phi) @ rho.unsqueeze(-1)).squeeze(-1)
    return torch.cat([t, so3_exp(phi)], dim=-1)


def se3_mul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """PyPose ``SE3_Mul``: a * b (apply b first)."""
    t = a[..., :3] + quat_act(a[..., 3:], b[..., :3])
    q = quat_mul(a[..., 3:], b[..., 3:])
    return torch.cat([t, q], dim=-1)


def se3_inv(a: torch.Tensor) -> torch.Tensor:
    qi = quat_inv(a[..., 3:])
    return torch.cat([-quat_act(qi, a[..., :3]), qi], dim=-1)


def se3_act(a: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    """``T.Act(p)`` = R p + t; ``a`` broadcasts against ``p`` [..., 3]."""
    return quat_act(a[..., 3:], p) + a[..., :3]


def se3_left_update(T: torch.Tensor, delta: torch.Tensor) -> torch.Tensor:
    """``LieTensor.add_`` as used by ``_Optimizer.update_parameter``: ``Exp(delta[:6]) * T``.

    The reference's step vector is 7 wide; the 7th entry is ignored
    (``Graphs.py:193-194`` "last column is useless").
    """
    return se3_mul(se3_exp(delta[..., :6]), T)


def so3_log(q: torch.Tensor) -> torch.Tensor:
    """Quaternion -> axis-angle (used only by tests to measure rotation error in rad)."""
    v, w = q[..., :3], q[..., 3:]
    sign = torch.where(w < 0, -torch.ones_like(w), torch.ones_like(w))
    v, w = v * sign, w * sign
    n = v.norm(dim=-1, keepdim=True)
    ang = 2.0 * torch.atan2(n, w)
    scale = torch.where(n > 1e-12, ang / torch.where(n > 1e-12, n, torch.ones_like(n)), 2.0 / w)
    return v * scale


def so3_Jl_inv(phi: torch.Tensor) -> torch.Tensor:
    """PyPose ``so3_Jl_inv``: I - K/2 + c K@K, c = (1 - theta cos(theta/2) / (2 sin(theta/2))) / theta^2."""
    eps = torch.finfo(phi.dtype).eps
    K = vec2skew(phi)
    theta = phi.norm(dim=-1, keepdim=True).unsqueeze(-1)
    safe = torch.where(theta > eps, theta, torch.ones_like(theta))
    half = 0.5 * safe
    c = torch.where(theta > eps, (1.0 - safe * half.cos() / (2.0 * half.sin())) / (safe * safe), 1.0 / 12.0 + theta * theta / 720.0)
    I = torch.eye(3, dtype=phi.dtype, device=phi.device).expand(K.shape)
    return I - 0.5 * K + c * (K @ K)


def se3_log(a: torch.Tensor) -> torch.Tensor:
    """PyPose ``SE3_Log``: SE3 [..., 7] -> [rho, phi] [..., 6] with phi = SO3_Log(q), rho = Jl^-1(phi) t."""
    phi = so3_log(a[..., 3:])
    rho = (so3_Jl_inv(phi) @ a[..., :3].unsqueeze(-1)).squeeze(-1)
    return torch.cat([rho, phi], dim=-1)


def pose_error(T_a: torch.Tensor, T_b: torch.Tensor) -> tuple[float, float]:
    """(translation error [m], rotation error [rad]) between two SE3 [7] poses."""
    d = se3_mul(se3_inv(T_a.double()), T_b.double())
    return float(d[..., :3].norm(dim=-1).max()), float(so3_log(d[..., 3:]).norm(dim=-1).max())
