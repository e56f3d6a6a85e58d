"""Seeded synthetic inputs shared by tests / bench / smoke.  CPU generator only.  (Moved here from tests/ in round 6: bench.py and
``__graft_entry__.smoke()`` no longer import workload generators from the test suite; ``tests/synth.py`` re-exports this module.)"""
import torch


def _exact_lognormal_like(shape, g, lo_exp=-4, n_exp=8):
    """Positive floats with a log-uniform-ish spread built from integer draws and exact fp32 arithmetic only
    ((1 + r1/1024) * 2^(r2 + lo_exp)), so that the same seed gives bit-identical tensors on any CPU (torch.randn /
    exp go through ISA-specific vector math and are NOT bit-reproducible across hosts)."""
    r1 = torch.randint(0, 1024, shape, generator=g).float()
    r2 = torch.randint(0, n_exp, shape, generator=g).float()
    return (1.0 + r1 / 1024.0) * torch.exp2(r2 + lo_exp)


def flow_cov_maps(H=480, W=640, seed=2, nan_frac=0.0):
    """S-sel: sigma_uu, sigma_vv spread over ~2 decades, sigma_uv = 0  -> [1,3,H,W] float32 (bit-reproducible)."""
    g = torch.Generator().manual_seed(seed)
    c = _exact_lognormal_like((1, 2, H, W), g)
    fc = torch.cat([c, torch.zeros(1, 1, H, W)], dim=1)
    if nan_frac > 0:
        m = torch.randint(0, 1 << 20, (1, 1, H, W), generator=g) < int(nan_frac * (1 << 20))
        fc[:, 0:1][m] = float("nan")
    return fc


def depth_maps(H=480, W=640, seed=3):
    """Tilted plane 3..58 m plus +-1/16 m quantised noise; depth cov = (z^2/80)^2 * spread * 1e-2 -> two [1,1,H,W]
    (integer draws + exact fp32 arithmetic: bit-reproducible across hosts)."""
    g = torch.Generator().manual_seed(seed)
    ys = (torch.arange(H, dtype=torch.int64)[:, None] * 40 * 1024 // H).float() / 1024.0
    xs = (torch.arange(W, dtype=torch.int64)[None, :] * 15 * 1024 // W).float() / 1024.0
    noise = (torch.randint(0, 129, (1, 1, H, W), generator=g).float() - 64.0) / 1024.0
    z = (3.0 + ys + xs)[None, None] + noise
    spread = _exact_lognormal_like((1, 1, H, W), g, lo_exp=-2, n_exp=3)
    zc = (z * z / 80.0) * (z * z / 80.0) * spread * 0.0078125
    return z.float(), zc.float()


def keypoints(n=200, H=480, W=640, seed=5, border=32):
    g = torch.Generator().manual_seed(seed)
    u = torch.randint(border, W - border, (n,), generator=g)
    v = torch.randint(border, H - border, (n,), generator=g)
    return torch.stack([u, v], dim=1)


# ----------------------------------------------------------------------------------------------------------
# S-e2e: a geometrically consistent synthetic stereo stream (planar scene, known SE3 trajectory)
# ----------------------------------------------------------------------------------------------------------
def coords_grid(B: int, H: int, W: int) -> torch.Tensor:
    """Pixel-coordinate grid ``[B, 2, H, W]`` (channel 0 = x, channel 1 = y): the zero-flow lookup coordinates of an H x W map."""
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    return torch.stack([xs, ys], dim=0).to(torch.float32)[None].repeat(B, 1, 1, 1)


class _se3:
    """The SE(3) arithmetic the generator needs (exponential map, composition, rotation matrix; poses are [tx, ty, tz, qx, qy, qz, qw], tangents
    [rho, phi]) — textbook closed forms, self-contained: the input generator does not import the oracle."""

    @staticmethod
    def _skew(v):
        z = torch.zeros_like(v[..., 0])
        return torch.stack([torch.stack([z, -v[..., 2], v[..., 1]], dim=-1), torch.stack([v[..., 2], z, -v[..., 0]], dim=-1),
                            torch.stack([-v[..., 1], v[..., 0], z], dim=-1)], dim=-2)

    @staticmethod
    def _so3_exp(phi):                                       # axis-angle -> unit quaternion (x, y, z, w)
        eps = torch.finfo(phi.dtype).eps
        theta = phi.norm(dim=-1, keepdim=True)
        theta2 = theta * theta
        theta4 = theta2 * theta2
        half = 0.5 * theta
        safe = torch.where(theta > eps, theta, torch.ones_like(theta))
        imag = torch.where(theta > eps, half.sin() / safe, 0.5 - theta2 / 48.0 + theta4 / 3840.0)
        real = torch.where(theta > eps, half.cos(), 1.0 - theta2 / 8.0 + theta4 / 384.0)
        return torch.cat([phi * imag, real], dim=-1)

    @staticmethod
    def _so3_Jl(phi):                                        # left Jacobian I + c1 K + c2 K^2
        eps = torch.finfo(phi.dtype).eps
        K = _se3._skew(phi)
        theta = phi.norm(dim=-1, keepdim=True).unsqueeze(-1)
        theta2 = theta * theta
        safe = torch.where(theta > eps, theta, torch.ones_like(theta))
        safe2 = safe * safe
        c1 = torch.where(theta > eps, (1.0 - safe.cos()) / safe2, 0.5 - theta2 / 24.0)
        c2 = torch.where(theta > eps, (safe - safe.sin()) / (safe * safe2), 1.0 / 6.0 - theta2 / 120.0)
        I = torch.eye(3, dtype=phi.dtype, device=phi.device).expand(K.shape)
        return I + c1 * K + c2 * (K @ K)

    @staticmethod
    def _quat_mul(a, b):
        av, aw, bv, bw = a[..., :3], a[..., 3:], b[..., :3], b[..., 3:]
        v = aw * bv + bw * av + torch.linalg.cross(av, bv)
        w = aw * bw - (av * bv).sum(dim=-1, keepdim=True)
        return torch.cat([v, w], dim=-1)

    @staticmethod
    def _quat_act(q, p):
        qv, qw = q[..., :3], q[..., 3:]
        uv = torch.linalg.cross(qv.expand(p.shape), p)
        uv = uv + uv
        return p + qw * uv + torch.linalg.cross(qv.expand(p.shape), uv)

    @staticmethod
    def quat_to_matrix(q):
        x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
        return torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], dim=-1),
                            torch.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], dim=-1),
                            torch.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=-1)], dim=-2)

    @staticmethod
    def se3_exp(xi):                                         # [rho, phi] -> [Jl(phi) rho, exp(phi)]
        rho, phi = xi[..., :3], xi[..., 3:6]
        t = (_se3._so3_Jl(phi) @ rho.unsqueeze(-1)).squeeze(-1)
        return torch.cat([t, _se3._so3_exp(phi)], dim=-1)

    @staticmethod
    def se3_mul(a, b):                                       # a * b (b first)
        t = a[..., :3] + _se3._quat_act(a[..., 3:], b[..., :3])
        return torch.cat([t, _se3._quat_mul(a[..., 3:], b[..., 3:])], dim=-1)

    @staticmethod
    def se3_inv(a):
        qi = torch.cat([-a[..., 3:6], a[..., 6:]], dim=-1)
        return torch.cat([-_se3._quat_act(qi, a[..., :3]), qi], dim=-1)

    @staticmethod
    def se3_act(a, p):                                       # R p + t
        return _se3._quat_act(a[..., 3:], p) + a[..., :3]


def _se3_exp(xi):
    return _se3.se3_exp(xi)


def make_camera(H=480, W=640):
    return dict(fx=320.0, fy=320.0, cx=W / 2.0, cy=H / 2.0, baseline=0.25, H=H, W=W)


def make_sequence(n_frames=4, H=480, W=640, C=256, iters=12, seed=0, feat_dtype=torch.float32, pool=2,
                  noise=0.3, device="cpu", closed_loop=False):
    """Returns (cam dict, list of frame dicts, list of true poses [7] float64).

    Frame dict keys = FrameInputs fields: fmap1, fmap2 [2,C,H/8,W/8]; coords [iters,2,2,H/8,W/8]; flow, logcov
    [2,2,H,W].  flow[0,0] = -disparity of the frame (stereo pair), flow[1] = temporal flow (t-1 -> t) sampled on
    frame t-1's pixel grid; both consistent with a planar scene and the true trajectory, plus noise ~ sigma.
    The feature maps / lookup coordinates are random (their consumer, the GRU, is not part of the hot path); only
    `pool` distinct sets are generated and cycled.
    """
    se3 = _se3

    dev = torch.device(device)
    g = torch.Generator().manual_seed(seed)
    cam = make_camera(H, W)
    fx, fy, cx, cy, bl = cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline"]
    h8, w8 = H // 8, W // 8
    if closed_loop:
        # closed trajectory (bench): circle of radius 0.3 m in the image plane + small periodic rotation, so that
        # frame 0 follows frame n-1 with the same inter-frame motion as every other pair
        import math

        poses = []
        for t in range(n_frames):
            th = 2 * math.pi * t / n_frames
            xi = torch.tensor([0.05 * math.sin(th), 0.3 * math.cos(th), 0.3 * math.sin(th),
                               0.01 * math.sin(th), 0.01 * math.cos(th), 0.01 * math.sin(2 * th)], dtype=torch.float64)
            poses.append(se3.se3_exp(xi))
    else:
        # trajectory: smooth forward motion with small rotations
        poses = [torch.tensor([0, 0, 0, 0, 0, 0, 1], dtype=torch.float64)]
        for _ in range(n_frames - 1):
            xi = torch.cat([torch.tensor([0.08, 0.0, 0.0]) + 0.02 * torch.randn(3, generator=g), 0.01 * torch.randn(3, generator=g)]).double()
            poses.append(se3.se3_mul(poses[-1], se3.se3_exp(xi)))
    # plane n . Pw = c in world NED (X forward): mostly fronto-parallel at ~12 m, tilted
    nrm = torch.tensor([1.0, 0.15, -0.25], dtype=torch.float64)
    nrm = nrm / nrm.norm()
    cpl = 12.0
    vs, us = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    dirs = torch.stack([torch.ones_like(us), (us - cx) / fx, (vs - cy) / fy], dim=-1)  # [H,W,3] NED ray, X = 1

    def depth_of(T):
        R = se3.quat_to_matrix(T[3:])
        rd = dirs @ R.T
        return (cpl - (nrm * T[:3]).sum()) / (rd @ nrm)  # [H,W] depth along X_cam

    pools = []
    for _ in range(pool):
        pools.append(dict(
            fmap1=torch.randn(2, C, h8, w8, generator=g).to(feat_dtype).to(dev),
            fmap2=torch.randn(2, C, h8, w8, generator=g).to(feat_dtype).to(dev),
            coords=(torch.stack([torch.arange(w8).float()[None].expand(h8, w8), torch.arange(h8).float()[:, None].expand(h8, w8)])[None, None]
                    + (torch.rand(iters, 2, 2, h8, w8, generator=g) * 2 - 1) * 8).to(dev),
        ))
    frames = []
    z_prev = depth_of(poses[-1]) if closed_loop else None
    for t in range(n_frames):
        T = poses[t]
        z = depth_of(T)
        disp = fx * bl / z
        logcov = (0.5 * torch.randn(2, 2, H, W, generator=g) - 0.7).float()
        sig = torch.exp(logcov)
        flow = torch.zeros(2, 2, H, W)
        flow[0, 0] = -(disp.float() + noise * sig[0, 0] * torch.randn(H, W, generator=g))
        flow[0, 1] = 0.01 * torch.randn(H, W, generator=g)
        if t > 0 or closed_loop:
            Tp = poses[t - 1]
            Pc = dirs * z_prev[..., None]                                   # points in camera t-1
            Pw = Pc @ se3.quat_to_matrix(Tp[3:]).T + Tp[:3]
            Rt = se3.quat_to_matrix(T[3:])
            P2 = (Pw - T[:3]) @ Rt                                           # R^T (Pw - t)
            u2 = fx * P2[..., 1] / P2[..., 0] + cx
            v2 = fy * P2[..., 2] / P2[..., 0] + cy
            flow[1, 0] = (u2 - us).float() + noise * sig[1, 0] * torch.randn(H, W, generator=g)
            flow[1, 1] = (v2 - vs).float() + noise * sig[1, 1] * torch.randn(H, W, generator=g)
        fr = dict(pools[t % pool])
        fr.update(flow=flow.to(dev), logcov=logcov.to(dev))
        frames.append(fr)
        z_prev = z
    return cam, frames, poses


def tartanair_sequence(C=32, iters=1):
    """The reference's own unit-test asset as hot-path inputs (tests/golden/tartanair_p000.npz, built by
    tests/golden/make_tartanair_fixture.py from Scripts/UnitTest/assets/test_sequence/TartanAir2_abs_P000): ground-truth
    depth -> disparity (stereo sample), ground-truth optical flow t-1 -> t (temporal sample), ground-truth NED poses.

    The log-sigma maps are synthetic and deterministic (integer arithmetic only): a small texture so that the 7x7 NMS has
    isolated minima, plus +3 where the flow is flagged invalid (occlusion / out of view) or the depth is sky, so the
    covariance-aware selector avoids those pixels exactly as it avoids high-uncertainty pixels of the real network.
    Returns (cam dict, frame dicts, poses [n,7] float64).
    """
    import os

    import numpy as np

    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tartanair_p000.npz"))
    depth = torch.from_numpy(z["depth"])                               # [n,H,W] f32
    flow = (torch.from_numpy(z["flow_u16"].astype(np.float32)) - 32768.0) / 64.0   # [n-1,2,H,W], exact
    fmask = torch.from_numpy(z["flow_mask"])                           # [n-1,H,W] 0 = valid
    poses = torch.from_numpy(z["poses"])
    fx, fy, cx, cy = [float(v) for v in z["K"]]
    bl = float(z["baseline"])
    n, H, W = depth.shape
    cam = dict(fx=fx, fy=fy, cx=cx, cy=cy, baseline=bl, H=H, W=W)
    g = torch.Generator().manual_seed(7)
    h8, w8 = H // 8, W // 8
    pool = dict(fmap1=torch.randn(2, C, h8, w8, generator=g), fmap2=torch.randn(2, C, h8, w8, generator=g),
                coords=(torch.stack([torch.arange(w8).float()[None].expand(h8, w8), torch.arange(h8).float()[:, None].expand(h8, w8)])[None, None]
                        + (torch.rand(iters, 2, 2, h8, w8, generator=g) * 2 - 1) * 4))
    vs, us = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    tex = (((us * 7 + vs * 13 + (us * vs) % 11) % 17).float() - 8.0) / 64.0        # exact multiples of 1/64 in [-1/8, 1/8]
    frames = []
    for t in range(n):
        fl = torch.zeros(2, 2, H, W)
        fl[0, 0] = -(fx * bl) / depth[t]                                # stereo sample: disparity (the frontend takes |.|)
        lc = torch.full((2, 2, H, W), -1.0) + tex
        lc[0] = lc[0] + 3.0 * (depth[t] > 100.0).float()
        if t > 0:
            fl[1] = flow[t - 1]
            bad = (fmask[t - 1] != 0) | (depth[t - 1] > 100.0)
            lc[1] = lc[1] + 3.0 * bad.float()
        fr = dict(pool)
        fr.update(flow=fl, logcov=lc)
        frames.append(fr)
    return cam, frames, poses


# ----------------------------------------------------------------------------------------------------------
# S-map: per-frame observation tables for the device-resident VisualMap tests (values are only copied, never computed on)
# ----------------------------------------------------------------------------------------------------------
def map_sequence(seed=21, n_frames=9, max_rows=60, lost_frames=(4, 5)):
    """Returns (meta, frames): meta = K [3,3], T_BS [7], baseline; frames[t] (t >= 1) holds one frame's tracking tables in the
    layout the HIP kernels leave them (SoA ``vals`` [11, n]) plus the validity mask, the prior / optimised pose and a
    timestamp.  Frames listed in ``lost_frames`` keep fewer than 10 rows (lost track -> need_interp)."""
    g = torch.Generator().manual_seed(seed)
    K = torch.tensor([[320.0, 0, 320.0], [0, 320.0, 240.0], [0, 0, 1.0]])
    T_BS = torch.tensor([0.1, -0.2, 0.05, 0.0, 0.0, 0.38268343, 0.92387953])
    frames = [dict(time_ns=1_700_000_000_000_000_000, n=0)]
    pose = torch.tensor([0.0, 0, 0, 0, 0, 0, 1])
    for t in range(1, n_frames):
        n = int(torch.randint(max_rows // 2, max_rows + 1, (1,), generator=g))
        valid = torch.rand(n, generator=g) > 0.15
        if t in lost_frames:
            valid[:] = False
            valid[: int(torch.randint(0, 9, (1,), generator=g))] = True
        A = torch.randn(n, 3, 3, generator=g, dtype=torch.float64)
        B = torch.randn(n, 3, 3, generator=g, dtype=torch.float64)
        q = torch.randn(4, generator=g) * 0.05 + torch.tensor([0, 0, 0, 1.0])
        opt = torch.cat([pose[:3] + torch.tensor([0.1, 0.01, -0.02]) + 0.01 * torch.randn(3, generator=g), q / q.norm()])
        frames.append(dict(
            n=n, valid=valid, time_ns=1_700_000_000_000_000_000 + t * 33_333_333,
            kp0=torch.randint(32, 600, (n, 2), generator=g).float(), kp1=torch.rand(n, 2, generator=g) * 500 + 40,
            vals=torch.rand(11, n, generator=g) * 20 + 0.5, sigma0=torch.tensor([0.25, 0.25, 0.0]).repeat(n, 1),
            sigma1=torch.rand(n, 3, generator=g) + 0.0625, cov0=A @ A.mT, cov1=B @ B.mT,
            pos_Tw=torch.randn(n, 3, generator=g) * 5, cov0w=B @ A @ A.mT @ B.mT,
            color=torch.randint(0, 256, (n, 3), generator=g, dtype=torch.uint8), prior=pose.clone(), opt=opt.float()))
        pose = opt.float()
    # dense-mapping tail (Odometry/MACVO.py:313-337): tracked frames also push map points (own generator: the tables above
    # stay what they were before this field existed)
    gm = torch.Generator().manual_seed(seed + 1000)
    for t in range(1, n_frames):
        fr = frames[t]
        nm = int(torch.randint(20, 51, (1,), generator=gm))
        C = torch.randn(nm, 3, 3, generator=gm, dtype=torch.float64)
        fr["map_pos_Tw"] = torch.randn(nm, 3, generator=gm) * 4
        fr["map_cov"] = C @ C.mT
        fr["map_color"] = torch.randint(0, 256, (nm, 3), generator=gm, dtype=torch.uint8)
    return dict(K=K, T_BS=T_BS, baseline=0.25), frames


def pgo_batch(probs, dev):
    """A list of pose-graph problems (objects with the fields of ``oracle.pgo.make_synthetic_problem``'s result) as one device-resident ``ops.PGOBatch``
    (concatenated per-point tables + the offsets table)."""
    from macvo_amd import ops

    off = [0]
    for p in probs:
        off.append(off[-1] + p.pos_Tw.shape[0])
    cat = lambda f: torch.cat([f(p) for p in probs]).contiguous().to(dev)  # noqa: E731
    return ops.PGOBatch(
        offsets=torch.tensor(off, dtype=torch.int32, device=dev),
        init_pose=torch.stack([p.init_pose for p in probs]).to(dev),
        intrinsics=torch.stack([torch.stack([p.K[0, 0], p.K[1, 1], p.K[0, 2], p.K[1, 2]]) for p in probs]).to(dev),
        baseline=torch.tensor([p.baseline for p in probs], dtype=torch.float32, device=dev),
        pos_Tw=cat(lambda p: p.pos_Tw), pixel2_uv=cat(lambda p: p.pixel2_uv), cov_Tw=cat(lambda p: p.cov_Tw),
        pixel2_d=cat(lambda p: p.pixel2_d.squeeze(-1)), pixel2_disp=cat(lambda p: p.pixel2_disp.squeeze(-1)),
        pixel2_disp_cov=cat(lambda p: p.pixel2_disp_cov.squeeze(-1)), pixel2_uv_cov=cat(lambda p: p.pixel2_uv_cov),
        obs2_covTc=cat(lambda p: p.obs2_covTc),
    )


# ----------------------------------------------------------------------------------------------------------
# generators of the measurement legs (bench.py kernels{} / patch_embed, tools/kernel_bench.py): the same values as the oracle's own
# generators (tests/test_abi_and_host.py pins the equality), without importing the oracle
# ----------------------------------------------------------------------------------------------------------
def patch_embed_weights(seed: int = 0, scale: float = 1.0):
    """Conv2d-default-like weights (uniform +-1/sqrt(fan_in)) of the three 6 x 6 stride-2 layers of the cost patch embedding: (w1, b1, w2, b2, w3, b3), fp32."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for cout, cin in ((16, 1), (32, 16), (64, 32)):
        bound = scale / (cin * 36) ** 0.5
        out.append((torch.rand(cout, cin, 6, 6, generator=g) * 2 - 1) * bound)
        out.append((torch.rand(cout, generator=g) * 2 - 1) * bound)
    return tuple(out)


def pgo_problem(n: int = 200, seed: int = 6, K=(320.0, 320.0, 320.0, 240.0), baseline: float = 0.25, W: int = 640, H: int = 480,
                trans_sigma: float = 0.1, rot_sigma: float = 0.02):
    """Seeded two-frame pose-graph problem ``(problem, T_true [7] float64)``: ``n`` points observed in camera 1 (identity prior) and, with pixel / disparity
    noise, in camera 2 at ``T_true``; ``problem`` carries the fields ``pgo_batch`` reads (NED points: depth first)."""
    from types import SimpleNamespace

    g = torch.Generator().manual_seed(seed)
    fx, fy, cx, cy = K
    Km = torch.tensor([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=torch.float32)
    depth = 2 + 18 * torch.rand(n, generator=g)
    u = 32 + (W - 64) * torch.rand(n, generator=g)
    v = 32 + (H - 64) * torch.rand(n, generator=g)
    p_w = torch.stack([depth, ((u - Km[0, 2]) * depth) / Km[0, 0], ((v - Km[1, 2]) * depth) / Km[1, 1]], dim=-1)      # world == camera-1 frame
    xi = torch.cat([trans_sigma * torch.randn(3, generator=g), rot_sigma * torch.randn(3, generator=g)]).double()
    T_true = _se3.se3_exp(xi)
    p_c2 = _se3.se3_act(_se3.se3_inv(T_true), p_w.double())
    Kd = Km.double()
    X, Y, Z = p_c2[..., 0], p_c2[..., 1], p_c2[..., 2]
    den = X.abs().clamp(min=torch.finfo(p_c2.dtype).tiny)
    den = torch.where(X >= 0, den, -den)
    uv2 = torch.stack([(Kd[0, 0] * Y + Kd[0, 2] * X) / den, (Kd[1, 1] * Z + Kd[1, 2] * X) / den], dim=-1)
    s_uu = torch.exp(2 * 0.5 * torch.randn(n, generator=g)).clamp(min=0.0625)
    s_vv = torch.exp(2 * 0.5 * torch.randn(n, generator=g)).clamp(min=0.0625)
    s_uv = torch.zeros(n)
    s_disp = (0.05 + 0.2 * torch.rand(n, generator=g))
    uv2n = uv2 + torch.stack([s_uu.sqrt() * torch.randn(n, generator=g), s_vv.sqrt() * torch.randn(n, generator=g)], -1) * 0.3
    disp2 = (fx * baseline) / p_c2[:, 0] + s_disp.sqrt() * torch.randn(n, generator=g) * 0.3
    d2 = (fx * baseline) / disp2
    sig_d = (0.02 * depth ** 2 / 20.0 + 0.05)
    A = torch.randn(n, 3, 3, generator=g, dtype=torch.float64) * 0.05
    cov_Tw = A @ A.transpose(-1, -2) + torch.diag_embed(torch.stack([sig_d, 0.01 * sig_d + 1e-3, 0.01 * sig_d + 1e-3], -1).double())
    Bm = torch.randn(n, 3, 3, generator=g, dtype=torch.float64) * 0.05
    obs_cov = Bm @ Bm.transpose(-1, -2) + torch.diag_embed(torch.stack([sig_d, 0.01 * sig_d + 1e-3, 0.01 * sig_d + 1e-3], -1).double())
    prob = SimpleNamespace(
        init_pose=torch.tensor([0, 0, 0, 0, 0, 0, 1], dtype=torch.float32), K=Km, baseline=baseline, pos_Tw=p_w.float(), cov_Tw=cov_Tw,
        pixel2_uv=uv2n.float(), pixel2_d=d2.float().unsqueeze(-1), pixel2_disp=disp2.float().unsqueeze(-1), pixel2_disp_cov=s_disp.float().unsqueeze(-1),
        pixel2_uv_cov=torch.stack([s_uu, s_vv, s_uv], -1).float(), obs2_covTc=obs_cov)
    return prob, T_true
