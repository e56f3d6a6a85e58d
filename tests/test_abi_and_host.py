"""CPU: the C-ABI library loads and exports every symbol include/macvo_hip.h declares; host-side logic; gloo gather."""
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "macvo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mv_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import macvo_amd._lib as L

    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    lib = L.load()
    decl = _declared_symbols()
    assert len(decl) >= 14
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in macvo_hip.h but not exported"
        assert name in L.SIGNATURES, f"{name} has no ctypes signature in _lib.py"
    assert sorted(L.SIGNATURES) == decl
    assert lib.mv_abi_version() == L.ABI_VERSION
    assert lib.mv_error_string(-4) == b"workspace too small"
    assert lib.mv_kp_select_workspace_bytes(480, 640) >= 2 * 480 * 640 * 4


def test_lm_defaults_match_reference_config():
    """init_context / _optimize constants (Optimizer.py:68-75,93)."""
    from macvo_amd import ops

    p = ops.lm_default_params()
    assert (p.huber_delta, p.radius, p.reject, p.max_steps, p.patience, p.decreasing) == (0.1, 1e3, 16, 10, 2, 1e-5)
    assert (p.diag_min, p.diag_max, p.pinv_rcond) == (1e-6, 1e32, 1e-15)


def test_ops_refuse_cpu_tensors():
    """no CPU fallback: the product path must fail loudly off-GPU"""
    from macvo_amd import _lib as L
    from macvo_amd import ops

    with pytest.raises(L.MacvoHipError):
        ops.corr_volume(torch.zeros(1, 16, 2, 2), torch.zeros(1, 16, 2, 2))


def test_product_package_does_not_import_oracle():
    out = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import macvo_amd, macvo_amd.ops, "
                          "macvo_amd.pipeline, macvo_amd.distributed, macvo_amd.plugins; "
                          "print(any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules))" % ROOT],
                         capture_output=True, text=True, check=True)
    assert out.stdout.strip() == "False"


def test_shard_sequences():
    from macvo_amd.distributed import shard_sequences

    owned = [shard_sequences(10, r, 4) for r in range(4)]
    assert sorted(sum(owned, [])) == list(range(10)) and owned[1] == [1, 5, 9]


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from macvo_amd.distributed import gather_poses, gather_tracks
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
T = 5 + rank                                   # ragged: rank r tracked 5 + r frames
poses = torch.full((T, 7), float(rank)) + torch.arange(T)[:, None]
stamps = (1 << 40) * (rank + 1) + torch.arange(T, dtype=torch.int64) * 33_333_333     # needs all 64 bits
out, ts, lengths = gather_tracks(poses, stamps, dist)
assert out.shape == (world, 5 + world - 1, 7), out.shape
assert lengths.tolist() == [5 + r for r in range(world)]
for r in range(world):
    n = int(lengths[r])
    assert torch.equal(out[r, :n], torch.full((n, 7), float(r)) + torch.arange(n)[:, None])
    assert torch.equal(ts[r, :n], (1 << 40) * (r + 1) + torch.arange(n, dtype=torch.int64) * 33_333_333)
    assert out[r, n:].abs().sum() == 0 and ts[r, n:].abs().sum() == 0
same, ln = gather_poses(torch.full((4, 7), float(rank)), dist)
assert same.shape == (world, 4, 7) and all(float(same[r].mean()) == r for r in range(world)) and ln.tolist() == [4] * world
dist.destroy_process_group()
print("OK", rank)
'''


def test_gloo_world2_pose_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29571", str(script)],
                         capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("OK") == 2


def test_missing_library_fails_loudly(tmp_path):
    """No CPU fallback: a missing libmacvo_hip.so must raise and name the build command (fresh interpreter: load() caches)."""
    import subprocess
    import sys

    code = ("import sys; sys.path.insert(0, %r); import macvo_amd; from macvo_amd import _lib as L\n"
            "try:\n    L.load(%r)\nexcept L.MacvoHipError as e:\n    assert 'no CPU fallback' in str(e) and 'make -C' in str(e); print('LOUD')\n"
            % (ROOT, str(tmp_path / "libmacvo_hip.so")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "LOUD" in out.stdout, out.stderr[-2000:]


def test_frame_pipe_config_validation_without_gpu():
    """mv_frame_pipe_arena_bytes is pure host code: sizes the arena and rejects invalid configurations with 0."""
    import ctypes as C

    from macvo_amd import _lib as L

    lib = L.load()
    lm = L.mvLMParams()
    lib.mv_lm_default_params(C.byref(lm))

    def cfg(**kw):
        d = dict(H=480, W=640, C=256, pairs=2, iters=12, radius=4, feat_dtype=L.MV_F32, layout=L.MV_LAYOUT_CHW, volume_split=0,
                 selector_mode=L.MV_KP_NODEPTH, kp_kernel_size=7, kp_mask_width=32, num_point=200, edgewidth=32,
                 min_num_point=10, graph_type=L.MV_GRAPH_DISP, filters=1, cov_kernel_size=31, fx=320.0, fy=320.0, cx=320.0,
                 cy=240.0, baseline=0.25, bl_fx=80.0, bl_fx_sq=6400.0, match_cov_default=0.25, max_match_cov=100.0,
                 max_depth_cov=250.0, max_depth=80.0, min_flow_cov_sq=0.0625, min_depth_cov=0.05, filter_min_depth=0.05,
                 map_max_depth=5.0, map_max_depth_cov=0.005, lm=lm)
        d.update(kw)
        return L.mvFramePipeConfig(**d)

    n = lib.mv_frame_pipe_arena_bytes(C.byref(cfg()))
    vol = 2 * 4800 * 4800 * 4
    assert 4 * vol < n < 4 * vol + 120e6 and n % 256 == 0            # one lane (round-5 layout): four volumes + ~75 MB of maps / token buffers / scratch
    assert lib.mv_frame_pipe_arena_bytes(C.byref(cfg(mapping=1, map_num_point=100, map_mask_width=32))) < 3 * vol + 120e6   # classic layout: three
    assert lib.mv_frame_pipe_default_depth(1, 0) == 3 and lib.mv_frame_pipe_default_depth(1, 1) == 2 and lib.mv_frame_pipe_default_depth(32, 0) == 3
    assert lib.mv_frame_pipe_arena_bytes(C.byref(cfg(H=720, W=1280))) > 3 * 2 * 14400 * 14400 * 4
    n32 = lib.mv_frame_pipe_arena_bytes(C.byref(cfg(pairs=64)))       # 32 lanes (batch-32 frames): two volumes of 64 pairs
    assert n32 > 3 * 32 * vol and n32 < 3 * 32 * vol + 32 * 100e6
    for bad in (dict(H=481), dict(C=100), dict(pairs=3), dict(pairs=0), dict(pairs=2 * L.MV_MAX_LANES + 2), dict(radius=5), dict(selector_mode=L.MV_KP_MAPPING),
                dict(graph_type=7), dict(volume_split=2), dict(volume_split=4, layout=L.MV_LAYOUT_HWC)):
        assert lib.mv_frame_pipe_arena_bytes(C.byref(cfg(**bad))) == 0, bad
    assert lib.mv_frame_pipe_arena_bytes(C.byref(cfg(volume_split=2, layout=L.MV_LAYOUT_HWC))) > n   # split planes added
    assert lib.mv_error_string(-4).decode() == "workspace too small"


def test_header_is_plain_c_and_layouts_match_ctypes(tmp_path):
    """gcc -std=c99 -pedantic compiles a C consumer of include/macvo_hip.h, links libmacvo_hip.so and runs host-only entry
    points; the struct sizes / offsets it prints must equal what the ctypes Structures in mac-vo_amd/_lib.py lay out."""
    import ctypes as C
    import shutil

    from macvo_amd import _lib as L

    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    L.load()
    exe = tmp_path / "abi_probe"
    libdir = os.path.join(ROOT, "mac-vo_amd")
    rocm = "/opt/rocm/lib"
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_abi", "abi_probe.c"), "-o", str(exe), "-L", libdir, "-lmacvo_hip",
           f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{rocm}", f"-Wl,-rpath-link,{rocm}"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    kv = {k.strip().split(" ")[-1] if k.startswith("sizeof") else k.strip(): v for k, v in re.findall(r"([A-Za-z_][\w ]*?)=(\S+)", out.stdout)}
    assert kv["abi"] == str(L.ABI_VERSION)
    assert int(kv["mvLMParams"]) == C.sizeof(L.mvLMParams)
    assert int(kv["mvFramePipeConfig"]) == C.sizeof(L.mvFramePipeConfig)
    assert int(kv["mvFrameInputs"]) == C.sizeof(L.mvFrameInputs)
    assert int(kv["mvKpSelectParams"]) == C.sizeof(L.mvKpSelectParams)
    assert int(kv["mvMatchCovParams"]) == C.sizeof(L.mvMatchCovParams)
    assert int(kv["offsetof lm"]) == L.mvFramePipeConfig.lm.offset and int(kv["fx"]) == L.mvFramePipeConfig.fx.offset
    assert int(kv["arena"]) > 2 * 4800 * 4800 * 4 * 2 and int(kv["MV_FB_POSE"]) == L.FB["POSE"] and int(kv["MV_BF16X2"]) == L.MV_BF16X2
    assert "steps=10" in out.stdout and "reject=16" in out.stdout and "err=unsupported" in out.stdout


def test_bench_gpus_n_launches_itself_and_gathers(tmp_path):
    """`python bench.py --gpus 2` with NO launcher in front of it (the form the round-end driver uses) must start one rank
    per GPU by itself and print rank 0's one JSON line.  `--dry-collectives` runs exactly that launch path on gloo / CPU tensors
    (rendezvous on 127.0.0.1, barrier, ragged gather_tracks, max-over-ranks clock) without touching a GPU."""
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--dry-collectives"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]           # ONE line, from rank 0
    import json

    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["gather_ok"] is True and line["dry_collectives"] is True
    assert [d["rank"] for d in line["rank_ids"]] == [0, 1] and line["track_lengths"] == [5, 5]
    assert line["value"] is None                       # a dry run can never be mistaken for a measurement


def test_native_permutation_generator_is_torch_randperm():
    """The frame driver's per-lane generator (std::mt19937 + partial Fisher-Yates + `discard` of the draws it skips,
    `mv_frame_pipe_finish_seeded`) against torch itself on the CPU: successive `torch.randperm(n, generator=g)[:k]` of one
    generator — ragged n from frame to frame, n < k, n = 0 / 1 / 2, 64-bit seeds (torch seeds its engine with the low 32 bits)."""
    import ctypes as C

    import numpy as np
    import torch

    from macvo_amd import _lib as L

    lib = L.load()
    for seed in (0, 1, 77, 123456789, 2 ** 40 + 5, 2 ** 63 + 11):
        ns = [8000, 7313, 200, 150, 1, 0, 2, 31, 100000, 4096, 9001]
        k = 200
        g = torch.Generator().manual_seed(seed if seed < 2 ** 63 else seed - 2 ** 64)
        want = [torch.randperm(n, generator=g)[:k] for n in ns]
        n_arr = np.asarray(ns, dtype=np.int64)
        out = np.full((len(ns), k), -1, dtype=np.int64)
        rc = lib.mv_randperm_heads(C.c_uint64(seed), n_arr.ctypes.data, len(ns), k, out.ctypes.data)
        assert rc == 0
        for i, (n, w) in enumerate(zip(ns, want)):
            m = min(k, n)
            assert np.array_equal(out[i, :m], w.numpy()), (seed, i, n)
            assert (out[i, m:] == -1).all()


def test_device_permutation_phases_are_torch_randperm():
    """Round 6: the DEVICE form of the same draw (csrc/randperm_dev.h: one-step block twist of the 624-word MT19937 state, partial Fisher-Yates restated as a
    hash lookup of `prev(i, p)` + pointer chains) — its phase functions executed on the host thread by thread (mv_randperm_heads_emulated, any emulated
    workgroup size) against torch.randperm itself.  The GPU suite runs the same functions as a kernel (tests/test_gpu_device_draw.py)."""
    import ctypes as C

    import numpy as np
    import torch

    from macvo_amd import _lib as L

    lib = L.load()
    assert lib.mv_randperm_state_words() >= 625 and lib.mv_randperm_max_head() >= 256
    ns = [8000, 3, 1, 0, 2, 200, 201, 199, 7000, 12345, 50, 100000, 624, 625, 623, 1248, 307200, 511, 512, 513]
    for seed in (0, 42, 2 ** 33 + 5):
        for k in (1, 100, 200, 512):
            for threads in (1, 64, 256, 1024):
                n_arr = np.asarray(ns, dtype=np.int64)
                out = np.full((len(ns), k), -1, dtype=np.int64)
                assert lib.mv_randperm_heads_emulated(C.c_uint64(seed), n_arr.ctypes.data, len(ns), k, threads, out.ctypes.data) == 0
                g = torch.Generator().manual_seed(seed)
                for i, n in enumerate(ns):
                    w = torch.randperm(n, generator=g)[:k].numpy()
                    assert np.array_equal(out[i, : len(w)], w), (seed, k, threads, i, n)
                    assert (out[i, len(w):] == -1).all()
    # the seeded device representation = init_genrand of the low 32 seed bits, position 624 (the first draw steps the block)
    st = np.zeros(lib.mv_randperm_state_words(), dtype=np.uint32)
    assert lib.mv_mt19937_seed(C.c_uint64(2 ** 40 + 5), st.ctypes.data) == 0
    assert st[0] == 5 and st[624] == 624 and st[1] == (1812433253 * (5 ^ (5 >> 30)) + 1) % 2 ** 32
    assert lib.mv_randperm_heads_emulated(C.c_uint64(0), n_arr.ctypes.data, 1, lib.mv_randperm_max_head() + 1, 64, out.ctypes.data) == -2   # MV_ERR_UNSUPPORTED

def test_rank_core_slices_are_numa_local_and_disjoint():
    """bench.py --gpus N pins each rank to its own host cores (round 6: NUMA-aware).  On the topology of the round's GPU boxes — 2 sockets x 64 cores x 2 threads, node 0 =
    cpus 0-63 + 128-191, node 1 = 64-127 + 192-255, GPUs 0-3 on node 0, 4-7 on node 1 — every rank gets physical cores of ITS GPU's node with their
    hyperthreads, the slices are disjoint and cover the machine; an unreadable topology falls back to the plain contiguous split."""
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod_cores", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    allowed = list(range(256))
    node = [list(range(0, 64)) + list(range(128, 192)), list(range(64, 128)) + list(range(192, 256))]
    gpu = lambda i: node[0] if i < 4 else node[1]  # noqa: E731
    sib = lambda c: (c % 128, c % 128 + 128)  # noqa: E731
    sl = [bench.rank_core_slice(r, 8, allowed, gpu, sib) for r in range(8)]
    assert all(len(x) == 32 for x in sl) and len(set().union(*map(set, sl))) == 256
    for r, x in enumerate(sl):
        assert set(x) <= set(node[0] if r < 4 else node[1])
        assert all((c % 128 + 128 in x) and (c % 128 in x) for c in x)          # whole physical cores
    assert sl[0] == list(range(0, 16)) + list(range(128, 144)) and sl[4] == list(range(64, 80)) + list(range(192, 208))
    # 2 ranks on a restricted mask, topology unknown -> plain split; one rank -> everything
    assert bench.rank_core_slice(1, 2, list(range(8)), lambda i: [], None) == [4, 5, 6, 7]
    assert bench.rank_core_slice(0, 1, list(range(8)), gpu, sib) == list(range(8))
    assert bench._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]

def test_bench_roofline_object_contract():
    """bench.py's `roofline` object (the driver's contract: bound / achieved / peak / unit / frac / traffic) for the three kinds of
    volume kernel, from synthetic launch times: achieved = algorithmic (or, for the split kernels, executed) work / time, frac =
    achieved / peak, SURVEY §8(d)'s per-unit work figures."""
    import importlib.util
    import types

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n_q, C = 4800, 256
    flops, nbytes = bench.volume_work(1, n_q, C, 4)
    assert flops == 2 * 2.0 * n_q * n_q * C and nbytes == 2 * (2.0 * n_q * C * 4 + 4.0 * n_q * n_q)
    ms = [0.1] * 200                                                   # 100 us per launch
    for kw, bound, peak, work in (
        (dict(feat_dtype="f32", layout="chw", volume_precision="exact"), "mfma", bench.PEAK_F32_MFMA_TFLOPS, flops / 1e12),
        (dict(feat_dtype="f32", layout="chw", volume_precision="f16x2"), "mfma", bench.PEAK_BF16_MFMA_TFLOPS, 3 * flops / 1e12),
        (dict(feat_dtype="f32", layout="chw", volume_precision="bf16x3"), "mfma", bench.PEAK_BF16_MFMA_TFLOPS, 6 * flops / 1e12),
        (dict(feat_dtype="f16", layout="hwc", volume_precision="exact"), "hbm", bench.PEAK_HBM_GBS, None),
    ):
        r = bench.roofline_of(ms, types.SimpleNamespace(**kw), 1, n_q, C, 20, traffic=123.0)
        assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us"} <= set(r)
        assert r["bound"] == bound and r["peak"] == peak and r["traffic"] == 123.0 and r["avg_launch_us"] == 100.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        if work is not None:
            assert abs(r["achieved"] - work / 1e-4) < 0.02 * r["achieved"] and r["unit"] == "TFLOP/s"
        else:
            b16 = bench.volume_work(1, n_q, C, 2)[1]
            assert abs(r["achieved"] - b16 / 1e-4 / 1e9) < 0.02 * r["achieved"] and r["unit"] == "GB/s"


def test_tiled_slice_cells_host_arithmetic():
    """`mv_tiled_slice_cells(H, W)` (include/macvo_hip.h): cells of one tiled fp16 volume slice = ceil(H / 4) tile rows of W / 4 tiles of 16 cells — the slice
    height padded to a multiple of 4; 0 where the tiling does not apply (W % 4 != 0).  Pure host arithmetic: callers size operand 2 and the volume with it."""
    from macvo_amd import _lib as L

    lib = L.load()
    assert lib.mv_tiled_slice_cells(60, 80) == 60 * 80                    # 640x480: no padding
    assert lib.mv_tiled_slice_cells(90, 160) == 92 * 160                  # 1280x720: 22.5 tile rows -> 23
    assert lib.mv_tiled_slice_cells(59, 64) == 60 * 64
    assert lib.mv_tiled_slice_cells(1, 4) == 16
    assert lib.mv_tiled_slice_cells(47, 156) == 48 * 156                  # KITTI-sized (1248x376) slices
    assert lib.mv_tiled_slice_cells(60, 78) == 0 and lib.mv_tiled_slice_cells(0, 80) == 0 and lib.mv_tiled_slice_cells(60, 0) == 0
    for h in range(1, 40):
        c = lib.mv_tiled_slice_cells(h, 8)
        assert c % 32 == 0 and h * 8 <= c < (h + 4) * 8


def test_measurement_generators_equal_the_oracle_generators():
    """`tools/synth.py` (input generators of bench.py / tools/kernel_bench.py; self-contained: the measurement legs do not import the oracle for their
    inputs) produces the oracle generators' values bit for bit: SE(3) helpers via a sequence, the two-frame pose-graph problem, the patch-embed weights,
    the coordinate grid."""
    import torch
    from oracle import corr, patch_embed, pgo, se3
    from tools import synth

    for n, seed in ((200, 6), (37, 1)):
        a, Ta = synth.pgo_problem(n, seed)
        b, Tb = pgo.make_synthetic_problem(n, seed)
        assert torch.equal(Ta, Tb) and a.baseline == b.baseline
        for f in ("init_pose", "K", "pos_Tw", "cov_Tw", "pixel2_uv", "pixel2_d", "pixel2_disp", "pixel2_disp_cov", "pixel2_uv_cov", "obs2_covTc"):
            assert torch.equal(getattr(a, f), getattr(b, f)), f
    for x, y in zip(synth.patch_embed_weights(3), patch_embed.make_weights(3)):
        assert torch.equal(x, y)
    assert torch.equal(synth.coords_grid(2, 5, 7), corr.coords_grid(2, 5, 7))
    xi = torch.randn(9, 6, dtype=torch.float64, generator=torch.Generator().manual_seed(2)) * 0.3
    xi[0] = 0
    T = synth._se3.se3_exp(xi)
    assert torch.equal(T, se3.se3_exp(xi)) and torch.equal(synth._se3.se3_mul(T, T.flip(0)), se3.se3_mul(T, T.flip(0)))
    assert torch.equal(synth._se3.se3_inv(T), se3.se3_inv(T)) and torch.equal(synth._se3.quat_to_matrix(T[:, 3:]), se3.quat_to_matrix(T[:, 3:]))
    import re

    assert not re.search(r"^\s*(from|import)\s+oracle\b", open(synth.__file__).read(), flags=re.M)      # the generator stands alone
