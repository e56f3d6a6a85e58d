"""Lane-batched frame driver (mv_frame_pipe_* with pairs = 2 * lanes; BASELINE configs[4] "batch-32 frames per GPU"),
free-running (non teacher-forced) sequence parity, and the full-size configurations through the frame driver
(BASELINE configs[2]: 1280x720, configs[4]: B = 64 pairs at 640x480)."""
import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu


def _inputs(fr, dev, **kw):
    from macvo_amd.pipeline import FrameInputs

    return FrameInputs(**{k: (None if v is None else v.to(dev)) for k, v in fr.items()}, **kw)


def _gens(seeds):
    return [torch.Generator().manual_seed(int(s)) for s in seeds]


@pytest.mark.parametrize("graph,selector", [("disp", "nodepth"), ("icp", "full")])
def test_lanes_equal_standalone_runs(gpu, graph, selector):
    """3 independent sequences advanced in lock-step through ONE pipe (one GEMM of 6 pairs, lane-batched small kernels, one
    batched LM solve) must reproduce, bit for bit, the three stand-alone single-lane runs (same kernels, same per-lane
    arithmetic, each lane drawing from a generator seeded like the stand-alone run)."""
    from macvo_amd.pipeline import Camera, HotPathConfig, NativeHotPath, stack_lanes

    H, W, n_frames, lanes = 240, 320, 6, 3
    seqs = [synth.make_sequence(n_frames, H, W, C=64, iters=3, seed=30 + l) for l in range(lanes)]
    cam = seqs[0][0]
    cfg = dict(graph_type=graph, selector=selector)
    solo = []
    for l in range(lanes):
        hp = NativeHotPath(Camera(**cam), HotPathConfig(**cfg), gpu, keep_extras=True, generators=_gens([500 + l]))
        ins = [_inputs(fr, gpu) for fr in seqs[l][1]]
        torch.cuda.synchronize()
        hp.initialize(ins[0])
        rec = []
        for t in range(1, n_frames):
            r = hp.step(ins[t])
            torch.cuda.synchronize()
            rec.append(dict(kp=r.kp0_uv.clone(), pose=r.pose.clone(), info=r.info.clone(), nv=r.n_valid.clone(),
                            cov0=r.extras["cov0"].clone(), cov1=r.extras["cov1"].clone(), valid=r.extras["valid"].clone(),
                            vals=r.extras["tracked"].vals.clone(), tok=hp.last_tokens.clone(), depth=hp.maps().depth.clone()))
        solo.append(rec)
    hp = NativeHotPath(Camera(**cam), HotPathConfig(**cfg), gpu, keep_extras=True, lanes=lanes,
                       generators=_gens([500 + l for l in range(lanes)]))
    batched = [stack_lanes([_inputs(seqs[l][1][t], gpu) for l in range(lanes)]) for t in range(n_frames)]
    torch.cuda.synchronize()
    hp.initialize(batched[0])
    for t in range(1, n_frames):
        res = hp.step(batched[t])
        torch.cuda.synchronize()
        assert len(res) == lanes
        for l, r in enumerate(res):
            ref = solo[l][t - 1]
            assert torch.equal(r.kp0_uv, ref["kp"]), (t, l)
            assert torch.equal(hp.last_tokens[2 * l: 2 * l + 2], ref["tok"])
            assert torch.equal(hp.maps(0, l).depth, ref["depth"])
            assert torch.equal(r.extras["tracked"].vals, ref["vals"])
            assert torch.equal(r.extras["cov0"], ref["cov0"]) and torch.equal(r.extras["cov1"], ref["cov1"])
            assert torch.equal(r.extras["valid"], ref["valid"]) and torch.equal(r.n_valid, ref["nv"])
            assert torch.equal(r.info, ref["info"]), (t, l, r.info, ref["info"])
            assert torch.equal(r.pose, ref["pose"]), (t, l)
    assert torch.equal(hp.pose, torch.stack([solo[l][-1]["pose"] for l in range(lanes)]))


def test_lanes_ragged_counts_and_lost_lane(gpu):
    """Lanes with different numbers of selected keypoints in one launch, one of them with NO candidates at all (its pose
    must stay at its prior, MACVO.py:303-307) while the others are optimised as if they ran alone."""
    from macvo_amd.pipeline import Camera, HotPathConfig, NativeHotPath, stack_lanes

    H, W, n_frames, lanes = 240, 320, 4, 3
    seqs = [synth.make_sequence(n_frames, H, W, C=64, iters=1, seed=60 + l) for l in range(lanes)]
    for fr in seqs[1][1]:                       # lane 1: huge flow covariance everywhere on the temporal pair -> quality above
        fr["logcov"] = fr["logcov"].clone()     # max_match_cov (100) -> no candidate survives
        fr["logcov"][1] = 4.0
    for fr in seqs[2][1]:                       # lane 2: only a small window has usable covariance -> fewer than 200 keypoints
        lc = fr["logcov"].clone()
        lc[1] = 4.0
        lc[1, :, 60:120, 60:160] = fr["logcov"][1, :, 60:120, 60:160]
        fr["logcov"] = lc
    cam = seqs[0][0]
    priors = torch.tensor([[0, 0, 0, 0, 0, 0, 1.0], [0.5, -0.25, 0.125, 0, 0, 0, 1.0], [0, 0, 0, 0, 0, 0, 1.0]])
    solo_pose, solo_nsel = [], []
    for l in range(lanes):
        hp = NativeHotPath(Camera(**cam), HotPathConfig(), gpu, generators=_gens([700 + l]))
        ins = [_inputs(fr, gpu) for fr in seqs[l][1]]
        torch.cuda.synchronize()
        hp.initialize(ins[0], init_pose=priors[l])
        ns = []
        for t in range(1, n_frames):
            ns.append(hp.step(ins[t]).n_sel)
        torch.cuda.synchronize()
        solo_pose.append(hp.pose.clone())
        solo_nsel.append(ns)
    assert all(n == 200 for n in solo_nsel[0]) and all(n == 0 for n in solo_nsel[1]) and all(0 < n < 200 for n in solo_nsel[2])
    hp = NativeHotPath(Camera(**cam), HotPathConfig(), gpu, lanes=lanes, generators=_gens([700 + l for l in range(lanes)]))
    batched = [stack_lanes([_inputs(seqs[l][1][t], gpu) for l in range(lanes)]) for t in range(n_frames)]
    torch.cuda.synchronize()
    hp.initialize(batched[0], init_pose=priors)
    sink = torch.zeros(n_frames - 1, lanes, 7, device=gpu)
    out = list(hp.run(batched[1:], pose_sink=sink))
    torch.cuda.synchronize()
    assert [[r.n_sel for r in res] for res in out] == [list(c) for c in zip(*solo_nsel)]
    assert torch.equal(hp.pose, torch.stack(solo_pose))
    assert torch.equal(hp.pose[1].cpu(), priors[1]) and torch.equal(sink[:, 1].cpu(), priors[1].expand(n_frames - 1, 7))
    assert out[-1][1].n_valid is None and out[-1][1].kp0_uv.shape == (0, 2)


@pytest.mark.parametrize("graph,driver", [("disp", "native"), ("reproj", "native"), ("icp", "python")])
def test_free_running_sequence_parity(gpu, graph, driver):
    """NO teacher forcing: 24 frames, each pipeline chained on its OWN previous pose.  Keypoints bit-exact on every frame,
    per-frame pose within the north_star tolerance (1e-4 m / 1e-4 rad) of the oracle, same LM step counts, and the relative
    translation error of the whole track (Evaluation/MetricsSeq.py:9-16, delta = 1) vs the oracle reported."""
    from macvo_amd.pipeline import Camera, HotPath, HotPathConfig, NativeHotPath
    from oracle import metrics, se3
    from oracle.pipeline import OracleHotPath

    n_frames = 25
    cam, frames, truth = synth.make_sequence(n_frames, 240, 320, C=32, iters=2, seed=41)
    ora = OracleHotPath(cam, dict(graph_type=graph))
    hot = (NativeHotPath if driver == "native" else HotPath)(Camera(**cam), HotPathConfig(graph_type=graph), gpu)
    ins = [_inputs(fr, gpu) for fr in frames]
    torch.cuda.synchronize()
    torch.manual_seed(77)
    ora.initialize(frames[0])
    ref = [ora.step(frames[t]) for t in range(1, n_frames)]
    torch.manual_seed(77)
    hot.initialize(ins[0])
    sink = torch.zeros(n_frames - 1, 7, device=gpu)
    steps, worst = [], (0.0, 0.0)
    for t, r in enumerate(hot.run(ins[1:], pose_sink=sink)):
        hot.sync_pose()
        assert torch.equal(r.kp0_uv.cpu(), ref[t]["kp0_uv"]), f"frame {t + 1}: keypoints differ"
        steps.append(int(r.info[0, 1].item()))
    torch.cuda.synchronize()
    est = torch.cat([torch.tensor([[0, 0, 0, 0, 0, 0, 1.0]]), sink.cpu()])
    orc = torch.cat([torch.tensor([[0, 0, 0, 0, 0, 0, 1.0]]), torch.stack([r["pose"] for r in ref])])
    for t in range(1, n_frames):
        dt, dr = se3.pose_error(orc[t].double(), est[t].double())
        worst = (max(worst[0], dt), max(worst[1], dr))
        assert dt <= 1e-4 and dr <= 1e-4, (t, dt, dr)
    assert steps == [r["steps"] for r in ref], (steps, [r["steps"] for r in ref])
    m = metrics.rte(orc, est)
    assert m["max"] <= 1e-4 and m["roe_max_rad"] <= 1e-4, m
    mt = metrics.rte(torch.stack(truth), est)
    assert mt["mean"] < 0.02, mt                     # sanity of the synthetic stream: the chained track follows the truth
    print(f"free-running {graph}/{driver}: worst per-frame diff {worst[0]:.2e} m / {worst[1]:.2e} rad; "
          f"RTE vs oracle mean {m['mean']:.2e} max {m['max']:.2e}; RTE vs truth mean {mt['mean']:.2e}")


def test_config2_720p_through_frame_driver(gpu):
    """BASELINE configs[2]: 1280x720 (N = 14400 queries, 829 MB volume per pair) through the native frame driver vs the
    oracle pipeline: tokens, keypoints (bit-exact), covariances, pose, for both the exact-fp32 features and the Fast-mode
    (bf16 features) volume — the latter checked against the oracle fed with the same bf16-rounded features."""
    from macvo_amd.pipeline import Camera, HotPathConfig, NativeHotPath
    from oracle import se3
    from oracle.pipeline import OracleHotPath

    from macvo_amd import ops

    H, W, n_frames = 720, 1280, 3
    cam, frames, _ = synth.make_sequence(n_frames, H, W, C=64, iters=2, seed=12)
    cam256, frames256, _ = synth.make_sequence(n_frames, H, W, C=256, iters=2, seed=12)
    # (dtype, layout, frames): fp32 / bf16 CHW at C = 64 (tile kernels), and the configuration of the 720p Fast-mode bench line:
    # f16 features, HWC, C = 256 -> the streaming kernel `corr_volume_h_stream`
    for dt, layout, frs, kernel in ((torch.float32, "chw", frames, None), (torch.bfloat16, "chw", frames, "corr_volume_h_chw"),
                                    (torch.float16, "hwc", frames256, "corr_volume_h_stream")):
        fr_dev = [dict(fr, fmap1=fr["fmap1"].to(dt), fmap2=fr["fmap2"].to(dt)) for fr in frs]
        if layout == "hwc":
            fr_dev = [dict(fr, fmap1=fr["fmap1"].permute(0, 2, 3, 1).contiguous(), fmap2=fr["fmap2"].permute(0, 2, 3, 1).contiguous())
                      for fr in fr_dev]
        fr_cpu = [dict(fr, fmap1=fr["fmap1"].to(dt).float(), fmap2=fr["fmap2"].to(dt).float()) for fr in frs]
        ora = OracleHotPath(cam, {})
        hot = NativeHotPath(Camera(**cam), HotPathConfig(feature_layout=layout), gpu, keep_extras=True)
        ins = [_inputs(fr, gpu) for fr in fr_dev]
        torch.cuda.synchronize()
        ora.initialize(fr_cpu[0])
        hot.initialize(ins[0])
        for t in range(1, n_frames):
            torch.manual_seed(900 + t)
            ro = ora.step(fr_cpu[t])
            torch.manual_seed(900 + t)
            rh = hot.step(ins[t])
            torch.cuda.synchronize()
            torch.testing.assert_close(hot.last_tokens.cpu(), ora.last_tokens, rtol=1e-5, atol=3e-4)
            assert torch.equal(rh.kp0_uv.cpu(), ro["kp0_uv"]), (dt, t)
            assert int(rh.n_valid.item()) == ro["n_valid"]
            inb = rh.extras["tracked"].inbound.cpu()
            torch.testing.assert_close(rh.extras["cov1"].cpu()[inb], ro["cov1"], rtol=2e-4, atol=1e-7)
            d_t, d_r = se3.pose_error(ro["pose"].double(), rh.pose.cpu().double())
            assert d_t <= 1e-4 and d_r <= 1e-4, (dt, t, d_t, d_r)
            assert int(rh.info[0, 1].item()) == ro["steps"]
        assert kernel is None or ops.last_volume_kernel() == kernel, (dt, layout, ops.last_volume_kernel())
        del hot
        torch.cuda.empty_cache()


def test_config4_batch32_through_frame_driver(gpu):
    """BASELINE configs[4]: batch-32 640x480 frames per GPU = 32 lanes = one volume GEMM of B = 64 pairs (5.9 GB) per step,
    through the native frame driver.  Every lane is checked bit for bit against a stand-alone single-lane run; lanes 0, 13
    and 31 additionally against the CPU oracle (keypoints bit-exact, pose <= 1e-4, LM step count)."""
    from macvo_amd.pipeline import Camera, HotPathConfig, NativeHotPath, stack_lanes
    from oracle import se3
    from oracle.pipeline import OracleHotPath

    H, W, n_frames, lanes, C, iters = 480, 640, 3, 32, 256, 4
    seqs = [synth.make_sequence(n_frames, H, W, C=C, iters=iters, seed=200 + l, pool=1) for l in range(lanes)]
    cam = seqs[0][0]
    hp = NativeHotPath(Camera(**cam), HotPathConfig(), gpu, lanes=lanes, generators=_gens([3000 + l for l in range(lanes)]))
    batched = [stack_lanes([_inputs(seqs[l][1][t], gpu) for l in range(lanes)]) for t in range(n_frames)]
    assert batched[0].fmap1.shape == (64, C, H // 8, W // 8)
    torch.cuda.synchronize()
    hp.initialize(batched[0])
    got = []
    for t in range(1, n_frames):
        res = hp.step(batched[t])
        torch.cuda.synchronize()
        got.append([(r.kp0_uv.clone(), r.pose.clone(), r.info.clone()) for r in res])
    toks = hp.last_tokens.clone()
    del hp, batched
    torch.cuda.empty_cache()
    for l in range(lanes):
        solo = NativeHotPath(Camera(**cam), HotPathConfig(), gpu, generators=_gens([3000 + l]))
        ins = [_inputs(fr, gpu) for fr in seqs[l][1]]
        torch.cuda.synchronize()
        solo.initialize(ins[0])
        for t in range(1, n_frames):
            r = solo.step(ins[t])
            torch.cuda.synchronize()
            kp, pose, info = got[t - 1][l]
            assert torch.equal(r.kp0_uv, kp) and torch.equal(r.pose, pose) and torch.equal(r.info, info), (l, t)
        assert torch.equal(solo.last_tokens, toks[2 * l: 2 * l + 2]), l
        del solo
    for l in (0, 13, 31):
        ora = OracleHotPath(cam, {})
        torch.manual_seed(3000 + l)
        ora.initialize(seqs[l][1][0])
        for t in range(1, n_frames):
            ro = ora.step(seqs[l][1][t])
            kp, pose, info = got[t - 1][l]
            assert torch.equal(kp.cpu(), ro["kp0_uv"]), (l, t)
            d_t, d_r = se3.pose_error(ro["pose"].double(), pose.cpu().double())
            assert d_t <= 1e-4 and d_r <= 1e-4, (l, t, d_t, d_r)
            assert int(info[0, 1].item()) == ro["steps"]


def test_native_seeded_lanes_equal_torch_generators(gpu):
    """`generators=[int, ...]`: the driver's own per-lane MT19937 + partial Fisher-Yates (mv_frame_pipe_seed_lanes /
    mv_frame_pipe_finish_seeded) must select exactly what `torch.Generator().manual_seed(seed)` + `torch.randperm(n)[:num]` selects —
    frame after frame (the generator state advances by n - 1 draws per call), with ragged candidate counts."""
    from macvo_amd.pipeline import Camera, HotPathConfig, NativeHotPath, stack_lanes

    H, W, n_frames, lanes = 240, 320, 6, 3
    seqs = [synth.make_sequence(n_frames, H, W, C=64, iters=2, seed=400 + l, pool=1) for l in range(lanes)]
    cam = seqs[0][0]
    seeds = [77, 2 ** 40 + 5, 123456789]                      # (torch seeds its MT19937 with the low 32 bits)
    a = NativeHotPath(Camera(**cam), HotPathConfig(num_point=150), gpu, lanes=lanes, generators=seeds)
    b = NativeHotPath(Camera(**cam), HotPathConfig(num_point=150), gpu, lanes=lanes, generators=_gens(seeds))
    batched = [stack_lanes([_inputs(seqs[l][1][t], gpu) for l in range(lanes)]) for t in range(n_frames)]
    torch.cuda.synchronize()
    a.initialize(batched[0])
    b.initialize(batched[0])
    sa, sb = torch.zeros(n_frames - 1, lanes, 7, device=gpu), torch.zeros(n_frames - 1, lanes, 7, device=gpu)
    ra = [[(r.kp0_uv.clone(), r.n_cand) for r in res] for res in a.run(batched[1:], pose_sink=sa) if a.sync_pose() is None]
    rb = [[(r.kp0_uv.clone(), r.n_cand) for r in res] for res in b.run(batched[1:], pose_sink=sb) if b.sync_pose() is None]
    torch.cuda.synchronize()
    assert len(ra) == n_frames - 1
    for t in range(n_frames - 1):
        for l in range(lanes):
            assert ra[t][l][1] == rb[t][l][1] and ra[t][l][1] > 150          # more candidates than selected points
            assert torch.equal(ra[t][l][0], rb[t][l][0]), (t, l)
    assert torch.equal(sa, sb)


@pytest.mark.parametrize("lanes,seeded,sync_each", [(1, True, False), (1, False, False), (2, True, True), (3, False, False)])
def test_backend_launch_thread_equals_inline_issue(gpu, lanes, seeded, sync_each):
    """`async_backend`: the launches of `finish` (+ the permutation draw of the seeded form) issued by the driver's second host thread
    while the caller's thread already enqueues the next frames.  Same bits as the inline form — every pose of a 3-deep pipelined run
    (device-side pose sink, no host sync in between when `sync_each` is off, so that the two threads really overlap), the newest
    frame's tables and LM info — and the same keypoints frame by frame when the consumer syncs after each frame (which drains the
    job queue: the other code path)."""
    from macvo_amd.pipeline import Camera, HotPathConfig, NativeHotPath, stack_lanes

    H, W, n_frames = 240, 320, 14
    seqs = [synth.make_sequence(n_frames, H, W, C=64, iters=2, seed=700 + l, pool=1) for l in range(lanes)]
    cam = seqs[0][0]
    seeds = [11 + 5 * l for l in range(lanes)]
    batched = [stack_lanes([_inputs(seqs[l][1][t], gpu) for l in range(lanes)]) for t in range(n_frames)]
    torch.cuda.synchronize()
    outs = []
    for mode in (True, False):
        hot = NativeHotPath(Camera(**cam), HotPathConfig(num_point=120, async_backend=mode), gpu, lanes=lanes,
                            generators=seeds if seeded else _gens(seeds), keep_extras=True)
        hot.initialize(batched[0])
        sink = torch.zeros(n_frames - 1, lanes, 7, device=gpu)
        per_frame = []
        for res in hot.run(batched[1:], pose_sink=sink):
            res = res if isinstance(res, list) else [res]
            if sync_each:
                hot.sync_pose()
                per_frame.append([(r.kp0_uv.clone(), r.pose.clone(), r.n_cand) for r in res])
            last = res
        hot.sync_pose()
        tail = [(r.kp0_uv.clone(), r.extras["tracked"].kp1_uv.clone(), r.extras["pos_Tw"].clone(), r.extras["cov0_w"].clone(),
                 r.extras["cov1"].clone(), r.info.clone(), r.pose_f64.clone(), r.pose.clone()) for r in last]
        torch.cuda.synchronize()
        outs.append((sink.clone(), per_frame, tail))
        del hot
    (sa, fa, ta), (sb, fb, tb) = outs
    assert sa.abs().sum().item() > 0
    assert torch.equal(sa, sb)
    for x, y in zip(ta, tb):
        for u, v in zip(x, y):
            assert torch.equal(u, v)
    assert len(fa) == len(fb)
    for x, y in zip(fa, fb):
        for (k0, p0, n0), (k1, p1, n1) in zip(x, y):
            assert n0 == n1 and torch.equal(k0, k1) and torch.equal(p0, p1)


def test_driver_tiled_volume_for_three_lanes_equals_row_major(gpu, monkeypatch):
    """`MV_PIPE_TILED=1` with a split volume precision: the frame driver packs operand 2 in tile order and runs
    `mv_corr_lookup_tiled` (measured slower in the pipeline, hence opt-in).  Same tokens, keypoints and poses as the
    same pipe with the row-major volume, bit for bit."""
    from macvo_amd.pipeline import Camera, HotPathConfig, NativeHotPath, stack_lanes

    H, W, n_frames, lanes = 480, 640, 4, 3
    seqs = [synth.make_sequence(n_frames, H, W, C=256, iters=3, seed=900 + l, pool=1) for l in range(lanes)]
    cam = seqs[0][0]
    batched = [stack_lanes([_inputs(seqs[l][1][t], gpu) for l in range(lanes)]) for t in range(n_frames)]
    torch.cuda.synchronize()
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("MV_PIPE_TILED", flag)
        hot = NativeHotPath(Camera(**cam), HotPathConfig(num_point=100, volume_precision="f16x2"), gpu, lanes=lanes,
                            generators=[5, 6, 7])
        hot.initialize(batched[0])
        sink = torch.zeros(n_frames - 1, lanes, 7, device=gpu)
        toks, kps = [], []
        for t in range(1, n_frames):
            res = hot.step(batched[t])
            toks.append(hot.last_tokens.clone())
            kps.append([r.kp0_uv.clone() for r in res])
            sink[t - 1] = torch.stack([r.pose for r in res])
        torch.cuda.synchronize()
        outs.append((toks, kps, sink.clone()))
        del hot
    for a, b in zip(outs[0][0], outs[1][0]):
        assert torch.equal(a, b)
    for fa, fb in zip(outs[0][1], outs[1][1]):
        for a, b in zip(fa, fb):
            assert torch.equal(a, b)
    assert torch.equal(outs[0][2], outs[1][2]) and outs[0][2].abs().sum().item() > 0
