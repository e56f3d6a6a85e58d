"""Round 6 — the device-driven frame: `selected_points[torch.randperm(n)[:numPoint]]` (Module/KeypointSelector.py:331,404) drawn ON THE GPU from a
device-resident MT19937 (csrc/randperm_dev.h), the candidate count never leaving device memory, `finish` never waiting for the GPU.  Parity bar: the same
bits as torch's CPU generator + torch.randperm — keypoint indices bit-exact — and, through the frame driver, the same tables and poses as the host-drawn
frame (which earlier rounds pinned against the oracle and the reference's own loop)."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu


def _inputs(frames, dev, **kw):
    from macvo_amd.pipeline import FrameInputs

    return [FrameInputs(**{k: (None if v is None else v.to(dev)) for k, v in fr.items()}, **kw) for fr in frames]


@pytest.mark.parametrize("k", [1, 200, 512])
def test_randperm_head_kernel_is_torch_randperm(gpu, k):
    """mv_randperm_head_lanes: one workgroup per lane, count read from device memory, generator advanced in place — successive calls of each lane's generator
    against `torch.Generator().manual_seed(seed)` + `torch.randperm(n)[:k]`, with counts around every edge (0, 1, 2, n <= k, block boundaries of the 624-word
    twist, a six-figure count)."""
    from macvo_amd import _lib as L

    lib = L.load()
    W = lib.mv_randperm_state_words()
    seeds = [0, 1, 1234, 2 ** 40 + 5]
    lanes = len(seeds)
    counts = [[8000, 3, 1, 0, 2, k, k + 1, max(k - 1, 0), 7000, 624, 625, 623, 1248, 100003, 50, 9999],
              [5, 0, 12345, 1, 1, 300, 8001, 200, 201, 199, 2, 0, 77777, 625, 8000, 8000],
              [307200, 1, 9000, 512, 513, 511, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7],
              [8000] * 16]
    st = np.zeros((lanes, W), dtype=np.uint32)
    for l, s in enumerate(seeds):
        L.check(lib.mv_mt19937_seed(C.c_uint64(s), st[l].ctypes.data), "mv_mt19937_seed")
    state = torch.from_numpy(st.view(np.int32)).to(gpu)
    gens = [torch.Generator().manual_seed(s) for s in seeds]
    cap = max(k, 1)
    for c in range(16):
        n = torch.tensor([[counts[l][c], -1, -1, -1] for l in range(lanes)], dtype=torch.int32, device=gpu)     # (count_stride = 4, as the selector writes it)
        out = torch.full((lanes, cap), -7, dtype=torch.int64, device=gpu)
        nsel = torch.full((lanes,), -1, dtype=torch.int32, device=gpu)
        L.check(lib.mv_randperm_head_lanes(state.data_ptr(), n.data_ptr(), 4, lanes, k, cap, out.data_ptr(), nsel.data_ptr(), 1, None), "mv_randperm_head_lanes")
        torch.cuda.synchronize()
        for l in range(lanes):
            want = torch.randperm(counts[l][c], generator=gens[l])[:k]
            assert int(nsel[l]) == want.numel(), (c, l)
            assert torch.equal(out[l, : want.numel()].cpu(), want), (c, l, counts[l][c])
            assert (out[l, want.numel():] == -7).all()


@pytest.mark.parametrize("lanes,selector,graph,num_point", [(1, "nodepth", "disp", 60), (1, "full", "icp", 200), (3, "nodepth", "reproj", 60), (1, "nodepth", "disp", 500)])
def test_device_driven_frame_equals_the_host_drawn_frame(gpu, monkeypatch, lanes, selector, graph, num_point):
    """The whole frame through the driver, device-driven (default for integer seeds) against `MV_PIPE_DEVICE_DRAW=0` (host MT19937 + kernel-argument / H2D
    permutation, the round-5 form pinned against torch, the oracle and the reference loop): the permutation head, EVERY backend table over its live rows, the
    counts and the poses are bit-identical, frame after frame (the generator advances by n - 1 draws per frame).  num_point = 500 > candidates covers the
    n <= k branch of the draw."""
    from macvo_amd.pipeline import Camera, HotPathConfig, NativeHotPath, stack_lanes

    H, W, Cc, n_frames = 192, 256, 32, 8
    cam, frames, _ = synth.make_sequence(n_frames + lanes, H, W, C=Cc, iters=2, seed=11)
    ins = _inputs(frames, gpu)
    batches = ins if lanes == 1 else [stack_lanes([ins[(t + l) % len(ins)] for l in range(lanes)]) for t in range(n_frames)]
    names = {"PERM": (torch.int64, ()), "KP0": (torch.int64, (2,)), "KP0F": (torch.float32, (2,)), "KP1": (torch.float32, (2,)), "INBOUND": (torch.uint8, ()),
             "SIGMA0": (torch.float32, (3,)), "SIGMA1": (torch.float32, (3,)), "POS_TC": (torch.float32, (3,)), "POS_TW": (torch.float32, (3,)),
             "COV0": (torch.float64, (9,)), "COV0W": (torch.float64, (9,)), "COV1": (torch.float64, (9,)), "VALID": (torch.uint8, ())}
    runs = {}
    for dd in ("0", "1"):
        monkeypatch.setenv("MV_PIPE_DEVICE_DRAW", dd)
        hot = NativeHotPath(Camera(**cam), HotPathConfig(num_point=num_point, graph_type=graph, selector=selector), gpu, lanes=lanes,
                            generators=[5 + 7 * l for l in range(lanes)])
        hot.initialize(batches[0])
        rec = []
        sink = torch.zeros((n_frames - 1, lanes, 7) if lanes > 1 else (n_frames - 1, 7), device=gpu)
        for t in range(1, n_frames):
            hot.enqueue_frontend(batches[t])
            res = hot.finish(None, sink[t - 1])
            assert hot.device_driven == (dd == "1")
            hot.sync_all()
            torch.cuda.synchronize()
            res = res if isinstance(res, list) else [res]
            cap = hot._cap
            d = {"n_sel": [r.n_sel for r in res], "n_cand": [r.n_cand for r in res]}
            for nm, (dt, tail) in names.items():
                if nm == "PERM" and dd == "0" and lanes == 1 and num_point <= 256:
                    continue                                   # (host-drawn, one lane: the permutation rides in the kernel arguments)
                v = hot._view(nm, 0, dt, (lanes, cap) + tail)
                d[nm] = [v[l, : res[l].n_sel].clone() for l in range(lanes)]
            vals = hot._view("VALS", 0, torch.float32, (11, lanes, cap))
            d["VALS"] = [vals[:, l, : res[l].n_sel].clone() for l in range(lanes)]
            for nm, dt, shp in (("ROT", torch.float64, (lanes, 9)), ("NVALID", torch.int32, (lanes,)), ("POSE64", torch.float64, (lanes, 7)),
                                ("INFO", torch.float64, (lanes, 4)), ("POSE", torch.float32, (lanes, 7))):
                d[nm] = [hot._view(nm, 0, dt, shp).clone()]
            rec.append(d)
        runs[dd] = (rec, sink.clone())
        del hot
    a, b = runs["0"], runs["1"]
    assert torch.equal(a[1], b[1]) and float(b[1].abs().sum()) > 0
    for t, (da, db) in enumerate(zip(a[0], b[0])):
        assert da["n_sel"] == db["n_sel"] and da["n_cand"] == db["n_cand"] and min(da["n_sel"]) > 0, (t, da["n_sel"], db["n_sel"])
        if num_point == 500:
            assert max(da["n_cand"]) <= 500                    # fewer candidates than num_point: every candidate is selected, in permuted order
        for k in db:
            if k in ("n_sel", "n_cand") or k not in da:
                continue
            for l, (x, y) in enumerate(zip(da[k], db[k])):
                assert torch.equal(x, y), (t, k, l)


def test_device_driven_stream_equals_torch_generators(gpu):
    """A pipelined stream (the driver's run loop: no host wait anywhere) of 3 ragged lanes, device-driven, against the same pipe drawing from
    `torch.Generator().manual_seed(seed)` on the Python side: keypoints, counts and poses equal frame by frame."""
    from macvo_amd.pipeline import Camera, FrameInputs, HotPathConfig, NativeHotPath, stack_lanes

    H, W, n_frames, lanes = 240, 320, 30, 3
    seqs = [synth.make_sequence(8, H, W, C=64, iters=2, seed=400 + l, pool=1, closed_loop=True) for l in range(lanes)]
    cam = seqs[0][0]
    seeds = [77, 2 ** 40 + 5, 123456789]
    one = lambda fr: FrameInputs(**{k: (None if v is None else v.to(gpu)) for k, v in fr.items()})  # noqa: E731
    pool = [stack_lanes([one(seqs[l][1][t]) for l in range(lanes)]) for t in range(8)]
    stream = [pool[t % 8] for t in range(n_frames)]
    outs = []
    for gens in (seeds, [torch.Generator().manual_seed(int(s)) for s in seeds]):
        hot = NativeHotPath(Camera(**cam), HotPathConfig(num_point=150), gpu, lanes=lanes, generators=gens)
        hot.initialize(stream[0])
        assert hot.device_driven == isinstance(gens[0], int)
        sink = torch.zeros(n_frames - 1, lanes, 7, device=gpu)
        rec = []
        for res in hot.run(stream[1:], pose_sink=sink):
            hot.sync_pose()
            rec.append([(r.kp0_uv.clone(), r.n_cand, r.n_sel) for r in res])
        torch.cuda.synchronize()
        outs.append((sink.clone(), rec))
        del hot
    (sa, ra), (sb, rb) = outs
    assert len(ra) == n_frames - 1 and torch.equal(sa, sb) and float(sa.abs().sum()) > 0
    for t in range(n_frames - 1):
        for l in range(lanes):
            assert ra[t][l][1:] == rb[t][l][1:] and ra[t][l][1] > 150, (t, l)
            assert torch.equal(ra[t][l][0], rb[t][l][0]), (t, l)


def test_device_driven_run_does_not_wait_for_the_gpu(gpu, monkeypatch):
    """What "device-driven" means: nothing in a frame waits for the host, so — with the run loop's flow control switched off (MV_PIPE_DD_AHEAD=-1; by default
    the host stays two finished frames ahead, profiles/r06_device_draw_ab.log) — the host can queue a whole stream while the GPU is still busy with its first
    frames.  A host-drawn pipe cannot: it waits for every frame's candidate count."""
    import time

    from macvo_amd.pipeline import Camera, HotPathConfig, NativeHotPath

    monkeypatch.setenv("MV_PIPE_DD_AHEAD", "-1")
    cam, frames, _ = synth.make_sequence(8, 480, 640, C=256, iters=12, seed=5, closed_loop=True)
    ins = _inputs(frames, gpu, static=True)
    hot = NativeHotPath(Camera(**cam), HotPathConfig(), gpu, generators=[3])
    hot.initialize(ins[0])
    for _ in hot.run(ins[1 + k % 7] for k in range(20)):
        pass
    torch.cuda.synchronize()
    assert hot.device_driven
    n = 80
    sink = torch.zeros(n, 7, device=gpu)
    t0 = time.perf_counter()
    it = hot.run((ins[1 + k % 7] for k in range(n)), pose_sink=sink)
    res = [next(it) for _ in range(n)]          # every frame enqueued and finished on the host ...
    t_host = time.perf_counter() - t0
    for _ in it:
        pass
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    assert len(res) == n and bool((sink.abs().sum(dim=1) > 0).all())
    assert t_host < 0.8 * t_all, (t_host, t_all)     # ... well before the GPU was through with them
    assert hot.host_frames >= n and hot.host_issue_s > 0


@pytest.mark.parametrize("lanes", [1, 2])
def test_front_launch_on_the_decoder_stream_equals_the_backend_stream_form(gpu, monkeypatch, lanes):
    """`MV_PIPE_FRONT_ON=decoder`: a device-driven frame's front launch (permutation draw + gathers + covariance models) rides behind the frame's own selector
    segment on its decoder-side stream; the backend stream carries the solves only.  The generator's state buffers and the previous frame's maps are ordered by
    the ring event of the previous front launch.  Keypoints, counts and poses of a pipelined stream are bit-identical to the default placement."""
    from macvo_amd.pipeline import Camera, HotPathConfig, NativeHotPath, stack_lanes

    n_pool, n_steps = 8, 120
    cam, frames, _ = synth.make_sequence(n_pool + lanes, 256, 384, C=256, iters=4, seed=31, closed_loop=True)
    ins = _inputs(frames, gpu, static=True)
    pool = ins[:n_pool] if lanes == 1 else [stack_lanes([ins[(t + l) % len(ins)] for l in range(lanes)]) for t in range(n_pool)]
    outs = {}
    for where in ("side", "decoder"):          # (decoder = the default)
        monkeypatch.setenv("MV_PIPE_FRONT_ON", where)
        hot = NativeHotPath(Camera(**cam), HotPathConfig(num_point=80), gpu, lanes=lanes, generators=[43 + l for l in range(lanes)])
        hot.initialize(pool[0])
        sink = torch.zeros((n_steps, 7) if lanes == 1 else (n_steps, lanes, 7), device=gpu)
        rec = []
        for r in hot.run((pool[(1 + k) % n_pool] for k in range(n_steps)), pose_sink=sink):
            hot.sync_pose()
            rr = r if isinstance(r, list) else [r]
            rec.append((torch.cat([x.kp0_uv for x in rr]).clone(), [x.n_cand for x in rr], [x.n_sel for x in rr]))
        torch.cuda.synchronize()
        assert hot.device_driven
        outs[where] = (sink.clone(), rec)
        hot.close()
    a, b = outs["side"], outs["decoder"]
    assert torch.isfinite(b[0]).all() and b[0].abs().sum() > 0 and torch.equal(a[0], b[0])
    for (ka, ca, sa), (kb, cb, sb) in zip(a[1], b[1]):
        assert ca == cb and sa == sb and min(sa) > 0 and torch.equal(ka, kb)


def test_host_permuted_finish_inside_a_device_driven_pipe_keeps_the_generator(gpu, monkeypatch):
    """`mv_frame_pipe_finish` (an explicit host permutation) stays legal in a device-driven pipe: the device generators are carried over unchanged to the state
    buffer the next device-driven finish reads.  Frames 1 and 3 device-drawn, frame 2 finished with the identity head: frames 1 and 3 select what
    `torch.Generator(seed)` selects when only those two frames draw from it."""
    from macvo_amd import ops
    from macvo_amd.pipeline import Camera, HotPathConfig, NativeHotPath

    cam, frames, _ = synth.make_sequence(5, 240, 320, C=64, iters=2, seed=77)
    ins = _inputs(frames, gpu)
    num = 50
    hot = NativeHotPath(Camera(**cam), HotPathConfig(num_point=num), gpu, generators=[9])
    hot.initialize(ins[0])
    lib, Cc = hot._lib, ops.C
    got = []
    for t in (1, 2, 3):
        hot.enqueue_frontend(ins[t])
        if t == 2:
            nc = (Cc.c_int32 * 1)()
            ops.L.check(lib.mv_frame_pipe_wait_candidates(hot._pipe, nc), "wait_candidates")
            perm = torch.arange(min(num, nc[0]), dtype=torch.int64)
            ns = (Cc.c_int32 * 1)(perm.numel())
            ops.L.check(lib.mv_frame_pipe_finish(hot._pipe, perm.data_ptr(), ns, None), "finish")
            hot._n_fin += 1
            hot.sync_all()
            torch.cuda.synchronize()
            n2 = nc[0]
            kp2 = hot._view("KP0", 0, torch.int64, (1, num, 2))[0, : perm.numel()].clone()
        else:
            r = hot.finish()
            hot.sync_all()
            torch.cuda.synchronize()
            got.append((r.kp0_uv.clone(), r.n_cand))
    assert hot.device_driven
    cand2 = hot._view("CAND", 1, torch.int32, (1, 240 * 320))[0, :num].clone()          # frame 2's candidate list (one enqueue back)
    hot.close()
    # reference: the same pipe host-drawn from torch generators, with the identity head on frame 2
    g = torch.Generator().manual_seed(9)
    monkeypatch.setenv("MV_PIPE_DEVICE_DRAW", "0")
    ref = NativeHotPath(Camera(**cam), HotPathConfig(num_point=num), gpu, generators=[g])
    ref.initialize(ins[0])
    want = []
    for t in (1, 2, 3):
        ref.enqueue_frontend(ins[t])
        if t == 2:
            nc = (Cc.c_int32 * 1)()
            ops.L.check(ref._lib.mv_frame_pipe_wait_candidates(ref._pipe, nc), "wait_candidates")
            assert nc[0] == n2
            perm = torch.arange(min(num, nc[0]), dtype=torch.int64)
            ns = (Cc.c_int32 * 1)(perm.numel())
            ops.L.check(ref._lib.mv_frame_pipe_release(ref._pipe, None), "release")
            ops.L.check(ref._lib.mv_frame_pipe_finish(ref._pipe, perm.data_ptr(), ns, None), "finish")
            ref._n_fin += 1
            ref.sync_all()
            torch.cuda.synchronize()
            assert torch.equal(ref._view("KP0", 0, torch.int64, (1, num, 2))[0, : perm.numel()], kp2)
        else:
            r = ref.finish()
            ref.sync_all()
            torch.cuda.synchronize()
            want.append((r.kp0_uv.clone(), r.n_cand))
    ref.close()
    assert kp2[:, 0].numel() == num and torch.equal(kp2[:, 1] * 320 + kp2[:, 0], cand2.long())      # the identity head = the first candidates, in order
    for (a, na), (b, nb) in zip(got, want):
        assert na == nb and na > num and torch.equal(a, b)
