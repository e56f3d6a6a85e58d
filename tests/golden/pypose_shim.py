"""Minimal stand-in for the ``pypose`` package, used ONLY by ``make_golden.py`` in the build container.

The reference's optimizer (``Module/Optimization``) imports PyPose (``requirements.txt:1``), which is not installable
here.  To still execute the reference's OWN in-tree code — the three residual graphs + analytic Jacobians
(``TwoFramePGO/Graphs.py``), ``LM_analytic.step`` (``PyposeOptimizers.py:136-194``) and the ``_optimize`` loop
(``TwoFramePGO/Optimizer.py:81-102``) — this module provides the handful of PyPose symbols they touch, restated from
the published PyPose 0.6.8 sources (pypose/lietensor/{lietensor,operation}.py, pypose/optim/{optimizer,kernel,
corrector,strategy,solver,scheduler}.py).  Golden vectors produced through it therefore pin the reference's in-tree
logic; the PyPose pieces themselves remain "parity unpinned" (they are this file's restatement).
"""
from __future__ import annotations

import sys
import types

import torch
from torch import nn
from torch.optim import Optimizer


# ----------------------------------------------------------------------------------------------- lietensor ops
def vec2skew(v):
    z = torch.zeros_like(v[..., 0])
    return torch.stack([torch.stack([z, -v[..., 2], v[..., 1]], -1), torch.stack([v[..., 2], z, -v[..., 0]], -1),
                        torch.stack([-v[..., 1], v[..., 0], z], -1)], -2)


def _so3_act(q, p):
    qv, qw = q[..., :3], q[..., 3:]
    shape = torch.broadcast_shapes(qv.shape, p.shape)
    qv, pb = qv.expand(shape), p.expand(shape)
    uv = torch.linalg.cross(qv, pb)
    uv = uv + uv
    return pb + qw * uv + torch.linalg.cross(qv, uv)


def _so3_mul(a, b):
    av, aw, bv, bw = a[..., :3], a[..., 3:], b[..., :3], b[..., 3:]
    return torch.cat([aw * bv + bw * av + torch.linalg.cross(av, bv), aw * bw - (av * bv).sum(-1, keepdim=True)], -1)


def _so3_exp(x):
    eps = torch.finfo(x.dtype).eps
    th = x.norm(dim=-1, keepdim=True)
    th2, half = th * th, 0.5 * th
    th4 = th2 * th2
    safe = torch.where(th > eps, th, torch.ones_like(th))
    imag = torch.where(th > eps, half.sin() / safe, 0.5 - th2 / 48 + th4 / 3840)
    real = torch.where(th > eps, half.cos(), 1 - th2 / 8 + th4 / 384)
    return torch.cat([x * imag, real], -1)


def _so3_Jl(x):
    eps = torch.finfo(x.dtype).eps
    K = vec2skew(x)
    th = x.norm(dim=-1, keepdim=True).unsqueeze(-1)
    th2 = th * th
    safe = torch.where(th > eps, th, torch.ones_like(th))
    c1 = torch.where(th > eps, (1 - safe.cos()) / (safe * safe), 0.5 - th2 / 24)
    c2 = torch.where(th > eps, (safe - safe.sin()) / (safe * safe * safe), 1.0 / 6 - th2 / 120)
    I = torch.eye(3, dtype=x.dtype, device=x.device).expand(K.shape)
    return I + c1 * K + c2 * (K @ K)


def _raw(t):
    return t.as_subclass(torch.Tensor) if isinstance(t, torch.Tensor) else t


class _Rot:
    def __init__(self, q):
        self.q = q

    def matrix(self):
        I = torch.eye(3, dtype=self.q.dtype, device=self.q.device)
        return _so3_act(self.q.unsqueeze(-2), I).transpose(-1, -2)


def _so3_log(q):
    """SO3_Log: factor = 2 atan(|v| / w) / |v|, small-|v| series 2/w - 2|v|^2/(3 w^3)."""
    v, w = q[..., :3], q[..., 3:]
    n = v.norm(dim=-1, keepdim=True)
    eps = torch.finfo(q.dtype).eps
    safe = torch.where(n > eps, n, torch.ones_like(n))
    factor = torch.where(n > eps, 2.0 * torch.atan(safe / w) / safe, 2.0 / w - 2.0 * n * n / (3.0 * w * w * w))
    return v * factor


def _so3_Jl_inv(x):
    eps = torch.finfo(x.dtype).eps
    K = vec2skew(x)
    th = x.norm(dim=-1, keepdim=True).unsqueeze(-1)
    safe = torch.where(th > eps, th, torch.ones_like(th))
    half = 0.5 * safe
    c = torch.where(th > eps, (1.0 - safe * half.cos() / (2.0 * half.sin())) / (safe * safe), 1.0 / 12.0 + th * th / 720.0)
    I = torch.eye(3, dtype=x.dtype, device=x.device).expand(K.shape)
    return I - 0.5 * K + c * (K @ K)


class _Se3Algebra(torch.Tensor):
    """se3 tangent vectors [rho, phi] (what SE3.Log() returns; only .ltype / .Exp() / scaling are used by the reference)."""
    ltype = "se3"

    def Exp(self):
        x = _raw(self)
        t = (_so3_Jl(x[..., 3:]) @ x[..., :3].unsqueeze(-1)).squeeze(-1)
        return LieTensor(torch.cat([t, _so3_exp(x[..., 3:])], -1))


class LieTensor(torch.Tensor):
    """SE3 only ([tx ty tz qx qy qz qw])."""
    ltype = "SE3"

    @staticmethod
    def __new__(cls, data, ltype=None):
        if ltype == "se3":
            return torch.Tensor._make_subclass(_Se3Algebra, _raw(data))
        return torch.Tensor._make_subclass(cls, _raw(data))

    def Log(self):
        d = _raw(self)
        phi = _so3_log(d[..., 3:])
        rho = (_so3_Jl_inv(phi) @ d[..., :3].unsqueeze(-1)).squeeze(-1)
        return torch.Tensor._make_subclass(_Se3Algebra, torch.cat([rho, phi], -1))

    def __init__(self, data=None, ltype=None):
        pass

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        """PyPose keeps LieTensor-ness through views / casts / indexing (also for Parameters, whose default
        __torch_function__ would strip the subclass): re-wrap float results whose last dim is still 7."""
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **(kwargs or {}))
        if isinstance(out, torch.Tensor) and not isinstance(out, LieTensor) and out.dim() >= 1 \
                and out.shape[-1] == 7 and out.is_floating_point():
            return torch.Tensor._make_subclass(LieTensor, out.detach() if out.requires_grad else out)
        return out

    def tensor(self):
        return _raw(self)

    def Inv(self):
        d = _raw(self)
        qi = torch.cat([-d[..., 3:6], d[..., 6:]], -1)
        return LieTensor(torch.cat([-_so3_act(qi, d[..., :3]), qi], -1))

    def Act(self, p):
        d = _raw(self)
        return _so3_act(d[..., 3:], _raw(p)) + d[..., :3]

    def rotation(self):
        return _Rot(_raw(self)[..., 3:])

    def __mul__(self, other):
        if isinstance(other, LieTensor):
            a, b = _raw(self), _raw(other)
            return LieTensor(torch.cat([a[..., :3] + _so3_act(a[..., 3:], b[..., :3]), _so3_mul(a[..., 3:], b[..., 3:])], -1))
        return self.Act(other)

    __matmul__ = __mul__

    def add_(self, other):
        """LieType.add_: input.copy_(Exp(other[..., :6]) * input)."""
        x = _raw(other)[..., :6].reshape(_raw(self).shape[:-1] + (6,))
        t = (_so3_Jl(x[..., 3:]) @ x[..., :3].unsqueeze(-1)).squeeze(-1)
        e = LieTensor(torch.cat([t, _so3_exp(x[..., 3:])], -1))
        new = _raw(e * LieTensor(_raw(self)))
        _raw(self).copy_(new)
        return self


class Parameter(LieTensor, nn.Parameter):
    def __new__(cls, data=None, requires_grad=True):
        return LieTensor._make_subclass(cls, _raw(data), requires_grad)

    def __init__(self, data=None, requires_grad=True):
        pass


def SE3(data):
    return LieTensor(data)


def identity_SE3(*lsize, dtype=None, device=None):
    """pp.identity_SE3(*lsize): identity transforms of shape lsize + (7,) (no batch dim when lsize is empty, which is what
    StaticMotionModel / MACVO.initialize rely on: Module/MotionModel.py:136-137, Odometry/MACVO.py:160)."""
    d = torch.zeros(tuple(lsize) + (7,), dtype=dtype or torch.float32, device=device)
    d[..., 6] = 1.0
    return LieTensor(d)


def cumops(input, dim, ops):
    """pp.cumops: inclusive scan y_k = x_0 (ops) x_1 (ops) ... (ops) x_k by doubling (Hillis-Steele), earlier operand on the
    LEFT — the order in which `pose[0] @ cumops(motions)` rebuilds a trajectory (Module/MapProcessor.py:73-74; the reference
    notes that 0.6.7 and 0.6.8 differ here, :72).  Restated from memory: parity unpinned."""
    assert dim == 0
    v = LieTensor(_raw(input).clone())
    L, i = v.shape[0], 1
    while i < L:
        idx = torch.arange(i, L)
        new = ops(LieTensor(_raw(v)[idx - i]), LieTensor(_raw(v)[idx]))
        _raw(v)[idx] = _raw(new)
        i *= 2
    return v


def pixel2point(pixels, depth, intrinsics):
    fx, fy, cx, cy = intrinsics[..., 0, 0], intrinsics[..., 1, 1], intrinsics[..., 0, 2], intrinsics[..., 1, 2]
    z = depth
    return torch.stack([((pixels[..., 0] - cx) * z) / fx, ((pixels[..., 1] - cy) * z) / fy, z], -1)


def point2pixel(points, intrinsics, extrinsics=None):
    h = points @ intrinsics.mT
    tiny = torch.finfo(h.dtype).tiny
    den = h[..., -1:].abs().clamp(min=tiny)
    den = torch.where(h[..., -1:] >= 0, den, -den)
    return h[..., :-1] / den


# ----------------------------------------------------------------------------------------------- optim
class Huber(nn.Module):
    def __init__(self, delta=1.0):
        super().__init__()
        self.delta, self.delta2 = delta, delta ** 2

    def forward(self, x):
        mask = x.sqrt() < self.delta
        out = torch.zeros_like(x)
        out[mask] = x[mask]
        out[~mask] = 2 * self.delta * x[~mask].sqrt() - self.delta2
        return out


class Trivial(nn.Module):
    def forward(self, x=None, R=None, J=None):
        return x if R is None else (R, J)


class FastTriggs(nn.Module):
    def __init__(self, kernel):
        super().__init__()
        self.func = lambda x: kernel(x).sum()

    @torch.no_grad()
    def forward(self, R, J):
        x = R.square().sum(-1, keepdim=True)
        with torch.enable_grad():
            s = torch.autograd.functional.jacobian(self.func, x).sqrt()
        sj = s.expand_as(R).reshape(-1, 1)
        return s * R, sj * J


class PINV(nn.Module):
    def forward(self, A, b):
        return torch.linalg.pinv(A) @ b


class Cholesky(nn.Module):
    def forward(self, A, b):
        return torch.cholesky_solve(b, torch.linalg.cholesky(A))


class TrustRegion:
    def __init__(self, radius=1e6, high=0.5, low=1e-3, up=2.0, down=0.5, factor=0.5, max=1e16, min=1e-6):
        self.min, self.max = min, max
        self.defaults = {"radius": radius, "high": high, "low": low, "up": up, "down": down, "factor": factor,
                         "damping": 1.0 / radius}
        self.down = down

    def update(self, pg, last, loss, J, D, R, *a, **k):
        JD = J @ D
        quality = (last - loss) / -((JD).mT @ (2 * R + JD)).squeeze()
        pg["radius"] = 1.0 / pg["damping"]
        if quality > pg["high"]:
            pg["radius"] = pg["up"] * pg["radius"]
            pg["down"] = self.down
        elif quality > pg["low"]:
            pg["down"] = self.down
        else:
            pg["radius"] = pg["radius"] * pg["down"]
            pg["down"] = pg["down"] * pg["factor"]
        pg["down"] = max(self.min, min(pg["down"], self.max))
        pg["radius"] = max(self.min, min(pg["radius"], self.max))
        pg["damping"] = 1.0 / pg["radius"]


class RobustModel(nn.Module):
    def __init__(self, model, kernel=None, auto=False):
        super().__init__()
        self.model = model
        self.kernel = [Trivial()] if kernel is None else kernel

    def model_forward(self, input):
        if isinstance(input, dict):
            return self.model(**input)
        if isinstance(input, tuple):
            return self.model(*input)
        return self.model(input)

    def forward(self, input, target=None):
        out = self.model_forward(input)
        outs = list(out) if isinstance(out, tuple) else [out]
        return [o if target is None else o - target for o in outs]

    def loss(self, input, target):
        res = self.forward(input, target)
        return sum(self.kernel[0](r.square().sum(-1)).sum() for r in res)


class _Optimizer(Optimizer):
    def update_parameter(self, params, step):
        steps = step.split([p.numel() for p in params if p.requires_grad])
        [p.add_(d.view(p.shape)) for p, d in zip(params, steps) if p.requires_grad]


class StopOnPlateau:
    # reaction to rejected LM steps — restated from memory of pypose/optim/scheduler.py (0.6.x):
    #     if hasattr(self.optimizer, 'reject_count'):
    #         if self.optimizer.reject_count > 0: self._continual = False   # "Maximum rejected steps reached"
    # => threshold 1.  The golden generator also runs the alternative reading (threshold = optimizer.reject, i.e. stop
    # only after the maximum number of rejections) and stores both; 0 disables the rule.
    STOP_ON_REJECT = 1
    last_instance = None    # the generator reads steps / reject_count of the scheduler `_optimize` created

    def __init__(self, optimizer, steps, patience=5, decreasing=1e-3, verbose=False):
        self.optimizer, self.max_steps, self.steps = optimizer, steps, 0
        self.patience, self.patience_count, self.decreasing = patience, 0, decreasing
        self._continual = True
        StopOnPlateau.last_instance = self
        self.reject_hist = []

    def continual(self):
        return self._continual

    def step(self, loss):
        self.steps += 1
        self.reject_hist.append(int(getattr(self.optimizer, "reject_count", 0)))
        if self.steps >= self.max_steps:
            self._continual = False
        if (self.optimizer.last - loss) < self.decreasing:
            self.patience_count += 1
        else:
            self.patience_count = 0
        if self.patience_count >= self.patience:
            self._continual = False
        thr = StopOnPlateau.STOP_ON_REJECT
        if thr > 0 and hasattr(self.optimizer, "reject_count") and self.optimizer.reject_count >= thr:
            self._continual = False


def modjac(*a, **k):
    raise NotImplementedError("autograd Jacobian path is not exercised by the golden generator")


class LM(_Optimizer):
    def __init__(self, *a, **k):
        raise NotImplementedError


class _Permissive(types.ModuleType):
    """Unknown attributes (only touched by reference code paths the generator never runs, e.g. dataset modules
    evaluated at import time) resolve to an inert factory returning an identity SE3."""
    __path__: list = []

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)

        class _Inert(LieTensor):
            def __new__(cls, *a, **kw):
                return LieTensor.__new__(LieTensor, torch.tensor([[0.0, 0, 0, 0, 0, 0, 1]]))

            def __class_getitem__(cls, item):
                return cls

        _Inert.__name__ = k
        return _Inert


def install():
    """Register the shim as ``pypose`` and its sub-modules in sys.modules."""
    pp = _Permissive("pypose")
    for n in ("LieTensor", "Parameter", "SE3", "identity_SE3", "vec2skew", "pixel2point", "point2pixel", "cumops"):
        setattr(pp, n, globals()[n])
    pp.SE3_type = types.SimpleNamespace(Act=lambda pose, p: LieTensor(_raw(pose)).Act(p))
    pp.from_matrix = lambda *a, **k: LieTensor(torch.tensor([[0.0, 0, 0, 0, 0, 0, 1]]))
    pp.Act = lambda pose, p: pose.Act(p)
    mods = {"pypose": pp}
    spec = {
        "pypose.optim": {"LM": LM},
        "pypose.optim.functional": {"modjac": modjac},
        "pypose.optim.strategy": {"TrustRegion": TrustRegion},
        "pypose.optim.solver": {"Cholesky": Cholesky, "PINV": PINV},
        "pypose.optim.corrector": {"FastTriggs": FastTriggs},
        "pypose.optim.optimizer": {"_Optimizer": _Optimizer, "Trivial": Trivial, "RobustModel": RobustModel},
        "pypose.optim.kernel": {"Huber": Huber},
        "pypose.optim.scheduler": {"StopOnPlateau": StopOnPlateau},
    }
    for name, attrs in spec.items():
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        mods[name] = m
    pp.optim = mods["pypose.optim"]
    for sub in ("functional", "strategy", "solver", "corrector", "optimizer", "kernel", "scheduler"):
        setattr(pp.optim, sub, mods["pypose.optim." + sub])
    sys.modules.update(mods)

    import importlib.abc
    import importlib.machinery

    class _SubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        """any other ``pypose.<x>`` import (e.g. pypose.module in MotionModel.py) -> inert permissive module"""

        def find_spec(self, name, path, target=None):
            if name.startswith("pypose.") and name not in sys.modules:
                return importlib.machinery.ModuleSpec(name, self, is_package=True)

        def create_module(self, spec):
            return _Permissive(spec.name)

        def exec_module(self, m):
            pass

    sys.meta_path.insert(0, _SubFinder())
    return pp
