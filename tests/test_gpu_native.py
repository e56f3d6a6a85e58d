"""The native frame driver (mv_frame_pipe_*, csrc/frame_pipe.hip) must reproduce the Python-sequenced HotPath bit for bit:
same kernels, same launch arguments, only the host side differs."""
import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu


def _inputs(frames, dev, hwc=False, **kw):
    from macvo_amd.pipeline import FrameInputs

    out = []
    for fr in frames:
        d = {k: (None if v is None else v.to(dev)) for k, v in fr.items()}
        if hwc:
            d["fmap1"] = d["fmap1"].permute(0, 2, 3, 1).contiguous()
            d["fmap2"] = d["fmap2"].permute(0, 2, 3, 1).contiguous()
        out.append(FrameInputs(**d, **kw))
    torch.cuda.synchronize()
    return out


def _pair(cam, cfg_kw, dev):
    from macvo_amd.pipeline import Camera, HotPath, HotPathConfig, NativeHotPath

    return (HotPath(Camera(**cam), HotPathConfig(**cfg_kw), dev, keep_extras=True),
            NativeHotPath(Camera(**cam), HotPathConfig(**cfg_kw), dev, keep_extras=True))


@pytest.mark.parametrize("H,W,graph,selector", [(480, 640, "disp", "nodepth"), (240, 320, "icp", "full"),
                                                (240, 320, "reproj", "nodepth")])
def test_native_step_equals_python_step(gpu, H, W, graph, selector):
    n_frames = 6
    cam, frames, _ = synth.make_sequence(n_frames, H, W, C=64, iters=3, seed=21)
    ins = _inputs(frames, gpu)
    py, nat = _pair(cam, dict(graph_type=graph, selector=selector), gpu)
    py.initialize(ins[0])
    nat.initialize(ins[0])
    for t in range(1, n_frames):
        torch.manual_seed(300 + t)
        a = py.step(ins[t])
        torch.manual_seed(300 + t)
        b = nat.step(ins[t])
        torch.cuda.synchronize()
        assert torch.equal(py.last_tokens, nat.last_tokens)
        ma, mb = py.maps_prev_for_next, nat.maps()
        for f in ("depth", "depth_cov", "disparity", "disparity_cov", "flow", "flow_cov"):
            assert torch.equal(getattr(ma, f), getattr(mb, f), ), f
        assert torch.equal(a.kp0_uv, b.kp0_uv), t
        assert torch.equal(a.n_valid, b.n_valid)
        for k in ("cov0", "cov0_w", "cov1", "valid", "pos_Tw"):
            assert torch.equal(a.extras[k], b.extras[k]), k
        for f in ("kp0_uv", "kp1_uv", "inbound", "vals", "sigma0", "sigma1"):
            assert torch.equal(getattr(a.extras["tracked"], f), getattr(b.extras["tracked"], f)), f
        assert torch.equal(a.pose_f64, b.pose_f64) and torch.equal(a.info, b.info)
        assert torch.equal(a.pose, b.pose), t


def test_native_pipelined_run_equals_python_run(gpu):
    n_pool, n_steps = 6, 25
    cam, frames, _ = synth.make_sequence(n_pool, 240, 320, C=64, iters=3, seed=11, closed_loop=True)
    ins = _inputs(frames, gpu, static=True)
    py, nat = _pair(cam, {}, gpu)
    outs = []
    for hp in (py, nat):
        hp.keep_extras = False
        hp.initialize(ins[0])
        sink = torch.zeros(n_steps, 7, device=gpu)
        torch.manual_seed(5)
        kps = []
        for r in hp.run((ins[(1 + k) % n_pool] for k in range(n_steps)), pose_sink=sink):
            hp.sync_pose()            # results live on the pipe's own streams: order the current stream after them
            kps.append(r.kp0_uv.clone())
        torch.cuda.synchronize()
        outs.append((sink.clone(), kps))
    assert torch.equal(outs[0][0], outs[1][0])
    assert len(outs[1][1]) == n_steps and all(torch.equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))
    assert outs[1][0].abs().sum() > 0


def test_native_upsample_path_and_split3(gpu):
    H, W, n_frames = 240, 320, 4
    cam, frames, _ = synth.make_sequence(n_frames, H, W, C=64, iters=2, seed=9)
    g = torch.Generator().manual_seed(2)
    for fr in frames:
        fr["flow8"] = torch.nn.functional.avg_pool2d(fr["flow"], 8) / 8.0
        fr["cov8"] = torch.nn.functional.avg_pool2d(fr["logcov"], 8) / 8.0
        fr["up_mask"] = torch.randn(2, 576, H // 8, W // 8, generator=g)
        fr["cov_mask"] = torch.randn(2, 576, H // 8, W // 8, generator=g) * 0.25
        fr["flow"] = None
        fr["logcov"] = None
    ins = _inputs(frames, gpu, hwc=True)
    py, nat = _pair(cam, dict(feature_layout="hwc", volume_precision="split3"), gpu)
    py.initialize(ins[0])
    nat.initialize(ins[0])
    for t in range(1, n_frames):
        torch.manual_seed(40 + t)
        a = py.step(ins[t])
        torch.manual_seed(40 + t)
        b = nat.step(ins[t])
        torch.cuda.synchronize()
        assert torch.equal(py.last_tokens, nat.last_tokens)
        assert torch.equal(a.kp0_uv, b.kp0_uv) and torch.equal(a.pose, b.pose)


def test_native_no_candidates_keeps_the_prior(gpu):
    from macvo_amd.pipeline import Camera, HotPathConfig, NativeHotPath

    cam, frames, _ = synth.make_sequence(3, 240, 320, C=64, iters=1, seed=4)
    ins = _inputs(frames, gpu)
    nat = NativeHotPath(Camera(**cam), HotPathConfig(max_match_cov=0.0), gpu)   # quality < 0 never holds: no candidates
    prior = torch.tensor([0.5, -0.25, 0.125, 0, 0, 0, 1.0])
    nat.initialize(ins[0], init_pose=prior)
    sink = torch.zeros(2, 7, device=gpu)
    res = list(nat.run(ins[1:], pose_sink=sink))
    torch.cuda.synchronize()
    assert [r.n_sel for r in res] == [0, 0] and [r.n_cand for r in res] == [0, 0]
    assert torch.equal(sink.cpu(), prior.expand(2, 7)) and torch.equal(nat.pose.cpu(), prior)
    assert res[-1].n_valid is None and res[-1].kp0_uv.shape == (0, 2)


def test_native_errors_are_loud(gpu):
    from macvo_amd import _lib as L
    from macvo_amd.pipeline import Camera, HotPathConfig, NativeHotPath

    cam, frames, _ = synth.make_sequence(5, 240, 320, C=64, iters=1, seed=4)
    ins = _inputs(frames, gpu)
    nat = NativeHotPath(Camera(**cam), HotPathConfig(), gpu)
    nat.initialize(ins[0])
    with pytest.raises(L.MacvoHipError):
        nat.finish()                              # nothing pending
    nat.enqueue_frontend(ins[1])
    nat.enqueue_frontend(ins[2])
    nat.enqueue_frontend(ins[3])
    with pytest.raises(L.MacvoHipError):
        nat.enqueue_frontend(ins[4])              # more than three tracked frames in flight
    with pytest.raises(L.MacvoHipError):
        nat.enqueue_volume(ins[4])
        nat.enqueue_volume(ins[4])                # at most one GEMM ahead of its frame
    r1 = nat.finish()
    nat.finish()
    nat.enqueue_frontend(ins[4])                  # completes the frame whose GEMM was issued ahead
    nat.finish()
    with pytest.raises(L.MacvoHipError):
        r1.kp0_uv                                 # recycled two finishes ago
    nat.synchronize()


def test_native_long_stream_is_race_free(gpu):
    """600 software-pipelined frames at full rate (slots rotate 200-300 times, four streams in flight): the native driver
    must reproduce the Python-sequenced poses bit for bit — a stale-slot or missing-event bug shows up as a mismatch."""
    n_pool, n_steps = 12, 600
    cam, frames, _ = synth.make_sequence(n_pool, 240, 320, C=64, iters=4, seed=17, closed_loop=True)
    ins = _inputs(frames, gpu, static=True)
    py, nat = _pair(cam, {}, gpu)
    sinks = []
    for hp in (py, nat):
        hp.keep_extras = False
        hp.initialize(ins[0])
        sink = torch.zeros(n_steps, 7, device=gpu)
        torch.manual_seed(9)
        for _ in hp.run((ins[(1 + k) % n_pool] for k in range(n_steps)), pose_sink=sink):
            pass
        torch.cuda.synchronize()
        sinks.append(sink.clone())
    assert torch.isfinite(sinks[1]).all()
    bad = (sinks[0] != sinks[1]).any(dim=1).nonzero().flatten()
    assert bad.numel() == 0, f"first mismatching frames: {bad[:10].tolist()}"


def test_native_measurement_hooks(gpu):
    """mv_frame_pipe_time_volume / _volume_times / _timeline: HIP events the driver records on its own streams (bench.py's
    roofline leg and the unprofiled timeline of DESIGN.md §5)."""
    n_pool = 6
    cam, frames, _ = synth.make_sequence(n_pool, 240, 320, C=64, iters=2, seed=3, closed_loop=True)
    ins = _inputs(frames, gpu, static=True)
    _, nat = _pair(cam, {}, gpu)
    nat.initialize(ins[0])
    torch.manual_seed(1)
    for _ in nat.run(ins[(1 + k) % n_pool] for k in range(4)):
        pass
    nat.time_volume(5)
    for _ in nat.run(ins[(5 + k) % n_pool] for k in range(8)):
        pass
    ms = nat.volume_times_ms()
    assert len(ms) == 5 and all(0.0 < t < 50.0 for t in ms)            # only the first 5 GEMMs after arming are timed
    tl = nat.timeline_ms()
    assert len(tl) == 5
    for i, (g0, g1, lk, sel) in enumerate(tl):
        assert g0 <= g1 <= lk <= sel, (i, g0, g1, lk, sel)                 # GEMM -> its lookups -> its selector
        assert abs((g1 - g0) - ms[i]) < 1e-3
    assert all(tl[i + 1][0] >= tl[i][1] for i in range(4))                 # GEMMs are serial on their stream
    nat.time_volume(0)
    for _ in nat.run(ins[(13 + k) % n_pool] for k in range(2)):
        pass
    assert nat.volume_times_ms() == []


def test_volume_view_and_empty_finish_pose_rotation(gpu):
    """(1) MV_FB_VOLUME must hand out the volume buffer of the asked frame (the `case` once fell through into the
    token view); with a GEMM issued ahead only age 0 is valid.  (2) a finish with no keypoints in any lane keeps the pose at the
    prior but still ROTATES the pose slots, so MV_FB_POSE age 1 is the previous finish's pose, not the one before it."""
    from macvo_amd import ops
    from macvo_amd.pipeline import Camera, HotPathConfig, NativeHotPath

    H, W, C = 192, 256, 32
    cam, frames, _ = synth.make_sequence(5, H, W, C=C, iters=2, seed=5)
    ins = _inputs(frames, gpu)
    n8 = (H // 8) * (W // 8)
    hot = NativeHotPath(Camera(**cam), HotPathConfig(num_point=40), gpu)
    hot.initialize(ins[0])
    torch.manual_seed(1)
    hot.step(ins[1])
    torch.cuda.synchronize()
    vol = hot._view("VOLUME", 0, torch.float32, (2 * n8, 1, H // 8, W // 8))
    want = ops.corr_volume(ins[1].fmap1, ins[1].fmap2)
    assert torch.equal(vol, want)                                   # frame 1's volume, not its tokens
    assert vol.data_ptr() != hot.last_tokens.data_ptr()
    prev = hot._view("VOLUME", 1, torch.float32, (2 * n8, 1, H // 8, W // 8))
    assert torch.equal(prev, ops.corr_volume(ins[0].fmap1, ins[0].fmap2)) and prev.data_ptr() != vol.data_ptr()
    hot.enqueue_volume(ins[2])                                      # a GEMM issued ahead rewrites the older buffer: age 1 is gone
    with pytest.raises(ops.L.MacvoHipError):
        hot._view("VOLUME", 1, torch.float32, (2 * n8, 1, H // 8, W // 8))
    hot.enqueue_frontend(ins[2])
    torch.manual_seed(2)
    r2 = hot.finish()
    hot.sync_all()
    torch.cuda.synchronize()
    pose2 = r2.pose.clone()
    # an empty finish: drive the C entry point with n_sel = 0 (what finish() does when the selector found no candidate)
    hot.enqueue_frontend(ins[3])
    L, lib = ops.L, hot._lib
    L.check(lib.mv_frame_pipe_wait_candidates(hot._pipe, hot._ncand), "wait")
    hot._nsel[0] = 0
    L.check(lib.mv_frame_pipe_release(hot._pipe, ops._stream()), "release")
    L.check(lib.mv_frame_pipe_finish(hot._pipe, None, hot._nsel, None), "finish")
    hot._n_fin += 1
    hot.sync_all()
    torch.cuda.synchronize()
    assert torch.equal(hot._view("POSE", 0, torch.float32, (1, 7))[0], pose2)      # stays at the prior (MACVO.py:303-307)
    assert torch.equal(hot._view("POSE", 1, torch.float32, (1, 7))[0], pose2)      # = frame 2's pose: the slots rotated
    hot.enqueue_frontend(ins[4])
    torch.manual_seed(3)
    r4 = hot.finish()
    hot.sync_all()
    torch.cuda.synchronize()
    assert torch.equal(hot._view("POSE", 1, torch.float32, (1, 7))[0], pose2) and not torch.equal(r4.pose, pose2)


_BF16X3_PIPE = r"""
import hashlib, sys, torch
sys.path.insert(0, sys.argv[1])
from tests import synth
from tests.test_gpu_native import _inputs
from macvo_amd import ops
from macvo_amd.pipeline import Camera, HotPathConfig, NativeHotPath
gpu = torch.device("cuda:0")
cam, frames, _ = synth.make_sequence(7, 256, 384, C=256, iters=3, seed=31)
ins = _inputs(frames, gpu)
nat = NativeHotPath(Camera(**cam), HotPathConfig(volume_precision=sys.argv[2]), gpu)
nat.initialize(ins[0])
torch.manual_seed(5)
sink = torch.zeros(6, 7, device=gpu)
h = hashlib.sha1()
for r in nat.run(ins[1:], pose_sink=sink):            # pipelined: packs and GEMMs of up to three frames in flight
    nat.sync_pose()
    h.update(r.kp0_uv.cpu().numpy().tobytes())
torch.cuda.synchronize()
h.update(sink.cpu().numpy().tobytes())
h.update(nat.last_tokens.cpu().numpy().tobytes())
print(ops.last_volume_kernel(), h.hexdigest())
"""


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2"])
def test_native_split_volume_equals_python_driver_and_oracle(gpu, mode):
    """volume_precision="bf16x3" through both drivers (C = 256: the packed streaming GEMM): the native pipe — pack on another
    stream beside the previous GEMM, two operand sets — must reproduce the Python-sequenced run bit for bit, frame by frame and
    pipelined, whatever stream the pack runs on (MV_PIPE_PACK_ON); keypoints equal the ORACLE's (fp32 einsum volume) and the pose
    stays within the north-star tolerance — the volume's 1e-5-level differences do not reach the backend through the tokens."""
    import os, subprocess, sys

    from macvo_amd import ops
    from oracle import se3
    from oracle.pipeline import OracleHotPath

    H, W, n_frames = 256, 384, 5                      # 32 x 48 = 1536 queries = 24 sub-tiles of 64 columns
    cam, frames, _ = synth.make_sequence(n_frames, H, W, C=256, iters=3, seed=30)
    ins = _inputs(frames, gpu)
    py, nat = _pair(cam, dict(volume_precision=mode), gpu)
    ora = OracleHotPath(cam, {})
    py.initialize(ins[0])
    nat.initialize(ins[0])
    ora.initialize(frames[0])
    for t in range(1, n_frames):
        torch.manual_seed(70 + t)
        a = py.step(ins[t])
        torch.manual_seed(70 + t)
        b = nat.step(ins[t])
        torch.cuda.synchronize()
        assert ops.last_volume_kernel() == f"corr_volume_split_stream<{mode}>"
        torch.manual_seed(70 + t)
        ro = ora.step(frames[t])
        assert torch.equal(py.last_tokens, nat.last_tokens)
        assert torch.equal(a.kp0_uv, b.kp0_uv) and torch.equal(a.pose, b.pose) and torch.equal(a.pose_f64, b.pose_f64)
        torch.testing.assert_close(nat.last_tokens.cpu(), ora.last_tokens, rtol=1e-5, atol=3e-4)
        assert torch.equal(b.kp0_uv.cpu(), ro["kp0_uv"])
        d_t, d_r = se3.pose_error(ro["pose"].double(), b.pose.cpu().double())
        assert d_t <= 1e-4 and d_r <= 1e-4, (t, d_t, d_r)
        py.pose = b.pose.clone()
        ora.pose = ro["pose"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for where in ("back", "vol", "main"):
        r = subprocess.run([sys.executable, "-c", _BF16X3_PIPE, root, mode], env=dict(os.environ, MV_PIPE_PACK_ON=where), capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.split())
    assert all(o[0] == f"corr_volume_split_stream<{mode}>" for o in outs) and len({o[1] for o in outs}) == 1, outs


@pytest.mark.parametrize("lanes,graph", [(1, "disp"), (1, "icp"), (3, "reproj")])
def test_fused_backend_equals_the_five_launch_form(gpu, monkeypatch, lanes, graph):
    """the backend of a frame as TWO launches — mv_backend_front_lanes (gather + track + back-projection + both covariance
    models + observation filters; the lane's last workgroup filters) and mv_pgo_solve_posed (rotation into the world frame as the solve's
    prologue, pose sink as its second output) — against the five-launch form (MV_PIPE_FUSE_BACKEND=0) on the same frames and seeds: EVERY
    backend table the pipe exposes is bit-identical over its live rows, frame after frame, and so are the poses."""
    from macvo_amd.pipeline import Camera, HotPathConfig, NativeHotPath, stack_lanes

    H, W, C, n_frames = 192, 256, 32, 7
    cam, frames, _ = synth.make_sequence(n_frames + lanes, H, W, C=C, iters=2, seed=11)
    ins = _inputs(frames, gpu)
    batches = ins if lanes == 1 else [stack_lanes([ins[(t + l) % len(ins)] for l in range(lanes)]) for t in range(n_frames)]
    names = {"KP0": (torch.int64, (2,)), "KP0F": (torch.float32, (2,)), "KP1": (torch.float32, (2,)), "INBOUND": (torch.uint8, ()),
             "SIGMA0": (torch.float32, (3,)), "SIGMA1": (torch.float32, (3,)), "POS_TC": (torch.float32, (3,)), "POS_TW": (torch.float32, (3,)),
             "COV0": (torch.float64, (9,)), "COV0W": (torch.float64, (9,)), "COV1": (torch.float64, (9,)), "VALID": (torch.uint8, ())}
    runs = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("MV_PIPE_FUSE_BACKEND", fuse)
        hot = NativeHotPath(Camera(**cam), HotPathConfig(num_point=60, graph_type=graph), gpu, lanes=lanes,
                            generators=None if lanes == 1 else [5 + l for l in range(lanes)])
        hot.initialize(batches[0])
        torch.manual_seed(3)
        rec = []
        sink = torch.zeros((n_frames - 1, lanes, 7) if lanes > 1 else (n_frames - 1, 7), device=gpu)
        for t in range(1, n_frames):
            hot.enqueue_frontend(batches[t])
            res = hot.finish(None, sink[t - 1])
            hot.sync_all()
            torch.cuda.synchronize()
            res = res if isinstance(res, list) else [res]
            cap = hot._cap
            d = {"n_sel": [r.n_sel for r in res]}
            for nm, (dt, tail) in names.items():
                v = hot._view(nm, 0, dt, (lanes, cap) + tail)
                d[nm] = [v[l, : res[l].n_sel].clone() for l in range(lanes)]
            vals = hot._view("VALS", 0, torch.float32, (11, lanes, cap))
            d["VALS"] = [vals[:, l, : res[l].n_sel].clone() for l in range(lanes)]
            for nm, dt, shp in (("ROT", torch.float64, (lanes, 9)), ("NVALID", torch.int32, (lanes,)), ("POSE64", torch.float64, (lanes, 7)),
                                ("INFO", torch.float64, (lanes, 4)), ("POSE", torch.float32, (lanes, 7))):
                d[nm] = [hot._view(nm, 0, dt, shp).clone()]
            rec.append(d)
        runs[fuse] = (rec, sink.clone())
        del hot
    a, b = runs["0"], runs["1"]
    assert torch.equal(a[1], b[1]) and float(b[1].abs().sum()) > 0          # the pose sink: the solve's second output == the D2D copy
    for t, (da, db) in enumerate(zip(a[0], b[0])):
        assert da["n_sel"] == db["n_sel"] and min(da["n_sel"]) > 0
        for k in da:
            if k == "n_sel":
                continue
            for l, (x, y) in enumerate(zip(da[k], db[k])):
                assert torch.equal(x, y), (t, k, l)


@pytest.mark.parametrize("lanes,selector", [(1, "nodepth"), (1, "full"), (2, "nodepth")])
def test_two_decoder_stream_layout_equals_the_classic_layout(gpu, monkeypatch, lanes, selector):
    """Round 5: one- and two-lane pipes run even / odd frames' decoder sides (12 lookups + selector segment) on two streams, backend + solve in order on
    one, the GEMM on 32 CUs fewer (`MV_PIPE_LAYOUT`, default `alt`).  Same kernels, same arguments: every pose and keypoint must equal the classic
    four-stream layout's bit for bit over a pipelined stream — with the NODEPTH selector (consecutive selector segments overlap, own workspace each) and with
    the FULL one (segments ordered by event: it reads the previous frame's depth)."""
    from macvo_amd.pipeline import Camera, HotPathConfig, NativeHotPath, stack_lanes

    n_pool, n_steps = 8, 120
    cam, frames, _ = synth.make_sequence(n_pool + lanes, 256, 384, C=256, iters=4, seed=23, closed_loop=True)   # 32 x 48 queries: the packed f16x2 GEMM
    ins = _inputs(frames, gpu, static=True)
    pool = ins[:n_pool] if lanes == 1 else [stack_lanes([ins[(t + l) % len(ins)] for l in range(lanes)]) for t in range(n_pool)]
    outs = {}
    for layout in ("classic", "alt"):
        monkeypatch.setenv("MV_PIPE_LAYOUT", layout)
        hot = NativeHotPath(Camera(**cam), HotPathConfig(selector=selector, num_point=80), gpu, lanes=lanes, generators=[41 + l for l in range(lanes)])
        assert hot._depth == (3 if layout == "alt" else 2)
        hot.initialize(pool[0])
        sink = torch.zeros((n_steps, 7) if lanes == 1 else (n_steps, lanes, 7), device=gpu)
        kps = []
        for r in hot.run((pool[(1 + k) % n_pool] for k in range(n_steps)), pose_sink=sink):
            hot.sync_pose()
            kps.append(torch.cat([x.kp0_uv for x in (r if isinstance(r, list) else [r])]).clone())
        torch.cuda.synchronize()
        outs[layout] = (sink.clone(), kps, hot.last_tokens.clone())
        del hot
    a, b = outs["classic"], outs["alt"]
    assert torch.isfinite(b[0]).all() and b[0].abs().sum() > 0
    bad = (a[0] != b[0]).reshape(n_steps, -1).any(dim=1).nonzero().flatten()
    assert bad.numel() == 0, f"first mismatching frames: {bad[:10].tolist()}"
    assert all(torch.equal(x, y) for x, y in zip(a[1], b[1])) and torch.equal(a[2], b[2])
