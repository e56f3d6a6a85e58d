"""Covariance-weighted two-frame pose-graph solve (TEST INFRASTRUCTURE).

Restates, in plain torch float64:

* the three residual graphs and their analytic Jacobians —
  ``Module/Optimization/TwoFramePGO/Graphs.py``: ``ICP_TwoframePGO`` :33-73,
  ``Reproj_TwoFramePGO`` :76-118, ``ReprojDisp_TwoFramePGO`` :121-148,
  ``Analytic_*`` :151-231 (buffers are built in float32 and the module is then cast to float64,
  ``Optimizer.py:84-85`` — reproduced);
* the outer loop of ``TwoFrame_PGO._optimize`` (``Optimizer.py:81-102``): dense
  ``block_diag(pinverse(cov_i))`` weight recomputed every outer iteration,
  ``StopOnPlateau(steps=10, patience=2, decreasing=1e-5)``;
* ``LM_analytic.step`` (``Module/Optimization/PyposeOptimizers.py:160-194``) line by line;
* the PyPose 0.6.8 pieces it calls (``Huber``, ``FastTriggs``, ``RobustModel.loss``,
  ``TrustRegion``, ``PINV``, ``StopOnPlateau``, SE3 ``add_``) restated from the published
  algorithm — pypose is not installable here, so these are "parity unpinned"
  (SURVEY.md §8 A21 / Appendix B.2).  Where memory of PyPose is ambiguous the behaviour is a
  named parameter of :class:`LMParams` so it can be flipped when PyPose is available.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import torch

from . import se3
from .frontend import pixel2point_NED, point2pixel_NED

GRAPH_ICP, GRAPH_REPROJ, GRAPH_DISP = 0, 1, 2
GRAPH_NAMES = {"icp": GRAPH_ICP, "reproj": GRAPH_REPROJ, "disp": GRAPH_DISP}


@dataclass
class LMParams:
    """Knobs of ``init_context`` (``Optimizer.py:68-75``) + ``_optimize`` (:88-93)."""
    huber_delta: float = 0.1          # Huber(delta=0.1) for kernel and corrector
    radius: float = 1e3               # TrustRegion(radius=1e3) -> damping0 = 1/radius
    tr_high: float = 0.5
    tr_low: float = 1e-3
    tr_up: float = 2.0
    tr_down: float = 0.5
    tr_factor: float = 0.5
    tr_min: float = 1e-6
    tr_max: float = 1e16
    diag_min: float = 1e-6            # LM_analytic(min=1e-6)
    diag_max: float = 1e32            # LM_analytic(max=1e32)
    reject: int = 16
    max_steps: int = 10               # StopOnPlateau(steps=10)
    patience: int = 2
    decreasing: float = 1e-5
    pinv_rcond: float = 1e-15         # torch.linalg.pinv default rcond for the PINV solver / weights
    # StopOnPlateau's reaction to rejected steps (PyPose is un-vendored => restated from memory, hence a knob):
    #   1  (default) PyPose 0.6.x scheduler.py: ``if optimizer.reject_count > 0: stop``
    #   16 (= reject) stop only when the inner loop exhausted its rejections;  0: rejections never stop the loop
    stop_on_reject: int = 1


@dataclass
class PGOProblem:
    """One ``GraphInput`` (``Graphs.py:11-21``) flattened to tensors (map dtypes per ``VisualMap.py:15-69``)."""
    init_pose: torch.Tensor          # [7] float32  (frames.pose of the frame to optimise)
    K: torch.Tensor                  # [3,3] float32
    baseline: float                  # float32 value
    pos_Tw: torch.Tensor             # [N,3] float32
    cov_Tw: torch.Tensor             # [N,3,3] float64
    pixel2_uv: torch.Tensor          # [N,2] float32
    pixel2_d: torch.Tensor           # [N,1] float32
    pixel2_disp: torch.Tensor        # [N,1] float32
    pixel2_disp_cov: torch.Tensor    # [N,1] float32
    pixel2_uv_cov: torch.Tensor      # [N,3] float32 (uu, vv, uv)
    obs2_covTc: torch.Tensor         # [N,3,3] float64


@dataclass
class PGOResult:
    pose: torch.Tensor               # [7] float64
    loss: float
    steps: int
    history: list = field(default_factory=list)
    reject_count: int = 0            # of the LAST LM step (what StopOnPlateau sees)


# --------------------------------------------------------------------------- PyPose pieces
def huber(x: torch.Tensor, delta: float) -> torch.Tensor:
    """``pp.optim.kernel.Huber.forward`` on squared norms x >= 0."""
    mask = x.sqrt() < delta
    out = torch.zeros_like(x)
    out[mask] = x[mask]
    out[~mask] = 2 * delta * x[~mask].sqrt() - delta ** 2
    return out


def fast_triggs(R: torch.Tensor, J: torch.Tensor, delta: float):
    """``pp.optim.corrector.FastTriggs(Huber(delta))``: s = sqrt(rho'(|r|^2)) per residual block."""
    x = R.square().sum(-1, keepdim=True)
    sx = x.sqrt()
    drho = torch.where(sx < delta, torch.ones_like(x), delta / torch.where(sx < delta, torch.ones_like(sx), sx))
    s = drho.sqrt()
    sj = s.expand_as(R).reshape(-1, 1)
    return s * R, sj * J


def robust_loss(R: torch.Tensor, delta: float) -> torch.Tensor:
    """``RobustModel.loss``: sum_blocks Huber(|r_block|^2) — unweighted, uncorrected residuals."""
    return huber(R.square().sum(-1), delta).sum()


def pinv_sym(A: torch.Tensor, rcond: float) -> torch.Tensor:
    return torch.linalg.pinv(A, rtol=rcond, hermitian=False)


# --------------------------------------------------------------------------- graphs
class _Graph:
    def __init__(self, prob: PGOProblem, graph_type: int):
        self.type = graph_type
        self.T = prob.init_pose.to(torch.float64).clone()
        K32 = prob.K.to(torch.float32)
        self.K = K32.double()
        self.pos_Tw = prob.pos_Tw.to(torch.float32).double()
        self.N = self.pos_Tw.size(0)
        if graph_type == GRAPH_ICP:
            # Graphs.py:49-55 — points_Tc built in float32, then module cast to double
            pts32 = pixel2point_NED(prob.pixel2_uv.float(), prob.pixel2_d.float().squeeze(-1), K32)
            self.points_Tc = pts32.double()
            self.obs_covTc = prob.obs2_covTc.double()
            self.pts_covTw = prob.cov_Tw.double()
        else:
            self.kp2 = prob.pixel2_uv.float().double()
            uvc = prob.pixel2_uv_cov.float()
            cov_kp2 = torch.empty((self.N, 2, 2))
            cov_kp2[:, 0, 0] = uvc[:, 0]
            cov_kp2[:, 1, 1] = uvc[:, 1]
            cov_kp2[:, 0, 1] = uvc[:, 2]
            cov_kp2[:, 1, 0] = uvc[:, 2]
            if graph_type == GRAPH_DISP:
                self.baseline = torch.tensor(prob.baseline, dtype=torch.float32).double()
                self.kp2_disp = prob.pixel2_disp.float().double()
                cov = torch.zeros((self.N, 3, 3))
                cov[:, :2, :2] = cov_kp2
                cov[:, 2, 2] = prob.pixel2_disp_cov.float().squeeze(-1)
                self.cov = cov.double()
            else:
                self.cov = cov_kp2.double()

    # residual blocks [N, r]
    def forward(self) -> torch.Tensor:
        if self.type == GRAPH_ICP:
            return se3.se3_act(self.T, self.points_Tc) - self.pos_Tw
        self.pos_Tc = se3.se3_act(se3.se3_inv(self.T), self.pos_Tw)
        reproj = point2pixel_NED(self.pos_Tc, self.K) - self.kp2
        if self.type == GRAPH_REPROJ:
            return reproj
        depth_err = self.pos_Tc[:, 0:1].reciprocal() * (self.K[0, 0] * self.baseline) - self.kp2_disp
        return torch.cat((reproj, depth_err), dim=-1)

    def covariance_array(self) -> torch.Tensor:
        if self.type == GRAPH_ICP:
            R = se3.quat_to_matrix(self.T[3:]).expand(self.N, 3, 3)
            return (R @ self.obs_covTc @ R.transpose(-2, -1)) + self.pts_covTw
        return self.cov

    def build_jacobian(self) -> torch.Tensor:
        """[r*N, 7]; last column is the dead PyPose column."""
        N = self.N
        if self.type == GRAPH_ICP:
            J = torch.zeros((N, 3, 7), dtype=torch.float64)
            J[..., 0:3] = torch.eye(3, dtype=torch.float64)
            J[..., 3:6] = -se3.vec2skew(se3.se3_act(self.T, self.points_Tc))
            return J.view(-1, 7)
        fx, fy = self.K[0, 0], self.K[1, 1]
        x, y, z = self.pos_Tc[:, 0], self.pos_Tc[:, 1], self.pos_Tc[:, 2]
        x2 = x ** 2
        Jh = torch.zeros(N, 2, 3, dtype=torch.float64)
        Jh[:, 0, 0] = -fx * y / x2
        Jh[:, 0, 1] = fx / x
        Jh[:, 1, 0] = -fy * z / x2
        Jh[:, 1, 2] = fy / x
        R_T = se3.quat_to_matrix(self.T[3:]).transpose(-2, -1)
        Jt = torch.zeros(N, 3, 7, dtype=torch.float64)
        Jt[..., :3] = -R_T
        Jt[..., 3:6] = R_T @ se3.vec2skew(self.pos_Tw)
        J_reproj = Jh @ Jt
        if self.type == GRAPH_REPROJ:
            return J_reproj.view(-1, 7)
        J_disp = (-(self.baseline * fx) / x2).view(-1, 1, 1) * Jt[:, 0:1, :]
        return torch.cat((J_reproj, J_disp), dim=1).view(-1, 7)


# --------------------------------------------------------------------------- LM
class _LMState:
    def __init__(self, p: LMParams):
        self.p = p
        self.damping = 1.0 / p.radius
        self.down = p.tr_down
        self.loss = None
        self.last = None
        self.reject_count = 0

    def trust_region_update(self, last, loss, J, D, R):
        """``pp.optim.strategy.TrustRegion.update`` (unweighted quality ratio on corrected R, J)."""
        p = self.p
        JD = J @ D
        quality = (last - loss) / -((JD).mT @ (2 * R + JD)).squeeze()
        radius = 1.0 / self.damping
        if quality > p.tr_high:
            radius = p.tr_up * radius
            self.down = p.tr_down
        elif quality > p.tr_low:
            self.down = p.tr_down
        else:
            radius = radius * self.down
            self.down = self.down * p.tr_factor
        self.down = max(p.tr_min, min(self.down, p.tr_max))
        radius = max(p.tr_min, min(radius, p.tr_max))
        self.damping = 1.0 / radius


def lm_step(graph: _Graph, st: _LMState, weight: torch.Tensor) -> torch.Tensor:
    """``LM_analytic.step`` (``PyposeOptimizers.py:160-194``)."""
    p = st.p
    R = graph.forward()
    J = graph.build_jacobian()
    st.last = st.loss = st.loss if st.loss is not None else robust_loss(R, p.huber_delta)
    st.reject_count = 0
    R, J = fast_triggs(R, J, p.huber_delta)
    J_T = J.mT @ weight
    A = J_T @ J
    A.diagonal().clamp_(p.diag_min, p.diag_max)
    while st.last <= st.loss:
        A.diagonal().add_(A.diagonal() * st.damping)
        D = pinv_sym(A, p.pinv_rcond) @ (-J_T @ R.view(-1, 1))
        T_prev = graph.T.clone()
        graph.T = se3.se3_left_update(graph.T, D.view(-1))
        st.loss = robust_loss(graph.forward(), p.huber_delta)
        st.trust_region_update(st.last, st.loss, J, D, R.view(-1, 1))
        if st.last < st.loss and st.reject_count < p.reject:
            # reference re-applies Exp(-D); numerically Exp(-D) Exp(D) T == T up to roundoff
            graph.T = se3.se3_left_update(graph.T, -D.view(-1))
            del T_prev
            st.loss, st.reject_count = st.last, st.reject_count + 1
        else:
            break
    return st.loss


def solve(prob: PGOProblem, graph_type: int | str = "disp", params: LMParams | None = None) -> PGOResult:
    """``TwoFrame_PGO._optimize`` (``Optimizer.py:81-102``) for one problem."""
    if isinstance(graph_type, str):
        graph_type = GRAPH_NAMES[graph_type]
    p = params or LMParams()
    g = _Graph(prob, graph_type)
    st = _LMState(p)
    steps, patience_count, continual = 0, 0, True
    hist = []
    while continual:
        cov = g.covariance_array()
        weight = torch.block_diag(*torch.linalg.pinv(cov, rtol=p.pinv_rcond))
        loss = lm_step(g, st, weight)
        # StopOnPlateau.step
        steps += 1
        if steps >= p.max_steps:
            continual = False
        if (st.last - loss) < p.decreasing:
            patience_count += 1
        else:
            patience_count = 0
        if patience_count >= p.patience:
            continual = False
        if p.stop_on_reject > 0 and st.reject_count >= p.stop_on_reject:
            continual = False
        hist.append((float(loss), st.reject_count, st.damping))
    return PGOResult(pose=g.T.clone(), loss=float(st.loss), steps=steps, history=hist, reject_count=st.reject_count)


# --------------------------------------------------------------------------- synthetic problems (SURVEY §8(d) S-pgo)
def make_synthetic_problem(n: int = 200, seed: int = 6, K=(320.0, 320.0, 320.0, 240.0), baseline: float = 0.25,
                           W: int = 640, H: int = 480, trans_sigma: float = 0.1, rot_sigma: float = 0.02,
                           outlier_frac: float = 0.0):
    """Seeded two-frame problem: returns ``(PGOProblem, T_true [7] float64)``.

    Points are observed in camera-1 (identity prior, StaticMotionModel), the true pose of camera-2 in
    the world is ``T_true``; pixel noise ~ N(0, Sigma_uv), disparity noise ~ N(0, sigma_disp).
    """
    g = torch.Generator().manual_seed(seed)
    fx, fy, cx, cy = K
    Km = torch.tensor([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=torch.float32)
    depth = 2 + 18 * torch.rand(n, generator=g)
    u = 32 + (W - 64) * torch.rand(n, generator=g)
    v = 32 + (H - 64) * torch.rand(n, generator=g)
    p_w = pixel2point_NED(torch.stack([u, v], -1), depth, Km)  # world == camera-1 frame
    xi = torch.cat([trans_sigma * torch.randn(3, generator=g), rot_sigma * torch.randn(3, generator=g)]).double()
    T_true = se3.se3_exp(xi)
    p_c2 = se3.se3_act(se3.se3_inv(T_true), p_w.double())
    uv2 = point2pixel_NED(p_c2, Km.double())
    s_uu = torch.exp(2 * 0.5 * torch.randn(n, generator=g)).clamp(min=0.0625)
    s_vv = torch.exp(2 * 0.5 * torch.randn(n, generator=g)).clamp(min=0.0625)
    s_uv = torch.zeros(n)
    s_disp = (0.05 + 0.2 * torch.rand(n, generator=g))
    uv2n = uv2 + torch.stack([s_uu.sqrt() * torch.randn(n, generator=g), s_vv.sqrt() * torch.randn(n, generator=g)], -1) * 0.3
    disp2 = (fx * baseline) / p_c2[:, 0] + s_disp.sqrt() * torch.randn(n, generator=g) * 0.3
    d2 = (fx * baseline) / disp2
    if outlier_frac > 0:
        k = int(n * outlier_frac)
        uv2n[:k] += 25 * torch.randn(k, 2, generator=g)
    sig_d = (0.02 * depth ** 2 / 20.0 + 0.05)
    A = torch.randn(n, 3, 3, generator=g, dtype=torch.float64) * 0.05
    cov_Tw = A @ A.transpose(-1, -2) + torch.diag_embed(torch.stack([sig_d, 0.01 * sig_d + 1e-3, 0.01 * sig_d + 1e-3], -1).double())
    B = torch.randn(n, 3, 3, generator=g, dtype=torch.float64) * 0.05
    obs_cov = B @ B.transpose(-1, -2) + torch.diag_embed(torch.stack([sig_d, 0.01 * sig_d + 1e-3, 0.01 * sig_d + 1e-3], -1).double())
    prob = PGOProblem(
        init_pose=torch.tensor([0, 0, 0, 0, 0, 0, 1], dtype=torch.float32),
        K=Km, baseline=baseline,
        pos_Tw=p_w.float(), cov_Tw=cov_Tw,
        pixel2_uv=uv2n.float(), pixel2_d=d2.float().unsqueeze(-1),
        pixel2_disp=disp2.float().unsqueeze(-1), pixel2_disp_cov=s_disp.float().unsqueeze(-1),
        pixel2_uv_cov=torch.stack([s_uu, s_vv, s_uv], -1).float(), obs2_covTc=obs_cov,
    )
    return prob, T_true
