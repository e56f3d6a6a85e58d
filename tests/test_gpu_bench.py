"""bench.py on the GPU: (1) launched the way the round-end driver launches N > 1 — under ``torch.distributed.run`` — at world size 1, so that
the RCCL process group, the barrier / all_reduce(MAX) / gather_tracks collectives and the rank bookkeeping of the N-rank path run on real
hardware in GPUTEST; (2) the line's parity block: the kernels the line times (volume rows vs fp64 einsum, every lookup's
tokens vs the oracle) and, when the byte-compiled reference tree is present, the reference's own MACVO loop on the same frames."""
import json
import os
import socket
import subprocess
import sys

import pytest

from tests import refrun

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUICK = ["--no-ramp", "--exact-steps", "0", "--config4-steps", "0", "--fast-mode-steps", "0", "--no-decoder-leg", "--pool", "6", "--end-to-end-frames", "0", "--plugin-frames", "0"]


def _line(p):
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.gpu
def test_bench_under_torch_distributed_run_world_1(gpu):
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "12", "--warmup", "3", "--no-cpu-baseline"] + QUICK
    d = _line(subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env))
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["rank_devices"] == [0]          # the process group was RCCL with one rank
    assert d["steps"] == 12 and d["warmup"] == 3 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["host_cores_per_rank"] >= 1
    r = d["roofline"]
    assert r["kernel"].startswith("corr_volume_split_stream<f16x2>")                        # the library's one default precision
    assert 0 < r["frac"] < 1 and r["launches_in_timed_region"] == 12


@pytest.mark.gpu
def test_bench_line_parity_covers_the_kernels_it_times(gpu):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "3", "--cpu-frames", "6", "--parity-frames", "6",
           "--reference-frames", "6"] + QUICK
    d = _line(subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT))
    par = d["parity"]
    assert par["keypoints_bit_exact_frames"] == par["frames"] and par["max_pose_dt_m"] <= 1e-4 and par["max_pose_dr_rad"] <= 1e-4
    vl = par["volume_and_lookups"]
    assert vl["volume_kernel"] == d["roofline"]["kernel"] and vl["volume_precision"] == d["config"]["volume_precision"] == "f16x2"
    assert vl["volume_rows_sampled"] >= 96 and vl["volume_max_abs_err"] <= vl["volume_abs_bar"] and vl["volume_max_err_rel_sum_abs"] <= 1e-6
    assert vl["lookup_launches"] == 3 * 12 and vl["within_bar"] and par["within_north_star"]
    cb = d["cpu_baseline"]
    if refrun.reference_root() is not None:
        vr = par["vs_reference_loop"]
        assert "error" not in vr, vr
        assert vr["stored_keypoints_bit_exact_frames"] == vr["frames_compared_keypoints"] >= 4
        assert vr["max_pose_dt_m"] <= 1e-4 and vr["max_pose_dr_rad"] <= 1e-4 and vr["within_north_star"]
        assert cb["kind"] == "reference" and cb["parts"]["reference_run_pair_s"] > 0
    else:
        assert cb["kind"] == "port"
    assert d["roofline"]["traffic_source"] is None or "NOT measured in this process" in d["roofline"]["traffic_source"]


@pytest.mark.gpu
def test_bench_two_ranks_sharing_the_gpu(gpu):
    """VERDICT r4 next #8 — what one GPU allows of the N > 1 path: ``bench.py --gpus 2 --share-gpu`` self-launches two ranks under
    ``torch.distributed.run`` exactly as the driver's ``--gpus N`` does (rendezvous, OMP_NUM_THREADS derivation, per-rank core pinning), both ranks
    run the real kernels on device 0, every collective of the timed region (barrier, gather_tracks of two device-resident tracks, all_reduce(MAX) of the
    clock) crosses a real process boundary over gloo.  Not a scaling measurement (the ranks share one GPU) and the line says so; RCCL itself has still
    only seen one rank (no multi-GPU box is available to the builder)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "12", "--warmup", "3", "--no-cpu-baseline"] + QUICK
    d = _line(subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT))
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["rank_devices"] == [0, 0] and d["share_gpu_test_mode"] is True
    assert d["rank_pose_tracks_finite"] == [True, True]                       # both ranks tracked their own sequence and both tracks arrived
    assert d["value"] > 0 and d["steps"] == 12 and d["scaling"] == "weak"
    sl = d["rank_core_slices"]
    ncores = len(os.sched_getaffinity(0))
    if ncores >= 4:                                                           # two distinct, disjoint core slices
        assert sl[0] is not None and sl[1] is not None and sl[0][1] < sl[1][0], sl
        assert d["host_cores_per_rank"] == ncores // 2
    assert "no scaling curve" in d["multi_gpu_note"]


@pytest.mark.gpu
def test_bench_eight_ranks_sharing_the_gpu(gpu):
    """VERDICT r5 next #7 — the 8-rank rehearsal one GPU allows: ``bench.py --gpus 8 --share-gpu`` = eight ranks under ``torch.distributed.run`` exactly as the
    driver launches ``--gpus 8`` (rendezvous on 127.0.0.1, per-rank core slices, OMP_NUM_THREADS), eight arenas and eight frame drivers on device 0, the
    timed region's collectives (barrier, two-phase gather_tracks of eight device tracks, all_reduce(MAX)) across eight real processes over gloo.  Asserts:
    eight disjoint core slices, eight finite tracks, one busy host thread per rank with its issue time per frame in the line.  NOT a scaling measurement."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--share-gpu", "--steps", "12", "--warmup", "3", "--no-cpu-baseline", "--no-kernel-events"] + QUICK
    d = _line(subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT))
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["rank_devices"] == [0] * 8 and d["share_gpu_test_mode"] is True
    assert d["rank_pose_tracks_finite"] == [True] * 8
    assert d["value"] > 0 and d["steps"] == 12 and d["scaling"] == "weak"
    ncores = len(os.sched_getaffinity(0))
    sl = d["rank_core_slices"]
    if ncores >= 16:
        assert all(x is not None for x in sl) and all(sl[i][1] < sl[i + 1][0] for i in range(7)), sl
        assert d["host_cores_per_rank"] == ncores // 8
    hu = d["rank_host_issue_us_per_frame"]
    assert len(hu) == 8 and all(h is not None and 0 < h < 5000 for h in hu), hu
    assert d["host"]["device_driven"] is True and d["host"]["host_threads"] == (2 if ncores // 8 >= 3 else 1)     # (the launch thread needs a core of its own)
    assert "no scaling curve" in d["multi_gpu_note"]


@pytest.mark.gpu
def test_bench_fast_mode_line_checks_the_fp16_cell_kernels(gpu):
    """with ``--feat-dtype f16 --layout hwc --volume-store encoder`` the pipe runs corr_volume_h_stream<out16> and the lookup on fp16 cells;
    the line's parity block must exercise THOSE kernels (the pipe's own fp16 volume buffer and token buffer), with the fp16 bar, not the fp32 ones."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "3", "--cpu-frames", "2", "--parity-frames", "2", "--reference-frames", "0",
           "--feat-dtype", "f16", "--layout", "hwc", "--volume-store", "encoder"] + QUICK
    d = _line(subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT))
    vl = d["parity"]["volume_and_lookups"]
    assert "error" not in vl, vl
    assert vl["volume_cell_dtype"] == "f16" and vl["volume_kernel"].startswith("corr_volume_h_stream<out16>"), vl
    assert vl["volume_rows_sampled"] >= 96 and vl["volume_within_bar"] and vl["within_bar"] and vl["pipe_tokens_checked"] == 3
    assert d["roofline"]["kernel"] == "corr_volume_h_stream<out16>" and d["roofline"]["bound"] == "hbm"


@pytest.mark.gpu
def test_bench_line_carries_a_roofline_per_kernel(gpu):
    """`kernels{}` — one entry per SURVEY §8(d) kernel, measured alone (HIP events, back to back) in the run itself, with the
    algorithmic bytes / FLOPs the fractions are computed from — and the patch-embedding leg with its Fast-mode (16-bit in / out) form."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "3", "--no-cpu-baseline", "--no-ramp", "--exact-steps", "0", "--config4-steps", "0",
           "--fast-mode-steps", "20", "--pool", "6", "--end-to-end-frames", "0", "--plugin-frames", "0"]
    d = _line(subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT))
    fm = d["fast_mode"]                                                  # the Fast-mode leg: fp16 features, tiled fp16-stored volume, one lane and 32
    assert "error" not in fm and fm["one_lane"]["value"] > 0 and fm["lanes_32"]["value"] > 0, fm
    k = d["kernels"]
    assert "error" not in k, k
    for name in ("volume", "lookup_B2", "lookup_B64", "volume_out16", "lookup_B2_vol16", "lookup_B2_vol16_tiled", "lookup_B64_vol16", "lookup_B64_vol16_tiled", "convex_upsample", "convex_upsample_bf16_mask", "patch_embed",
                 "patch_embed_fast_mode", "selector", "covariance", "solve"):
        assert name in k and "error" not in k[name], (name, k.get(name))
        assert k[name]["us"] > 0
        if k[name].get("frac") is not None:
            assert 0 < k[name]["frac"] < 1, (name, k[name])
    assert k["convex_upsample"]["algorithmic_bytes"] == 2 * 4800 * (576 * 4 + 8 + 512) and k["convex_upsample"]["us"] < 15          # (r4: 23-26 us)
    assert k["lookup_B2"]["algorithmic_bytes"] == 2 * (4800 * 100 * 4 + 4800 * 8 + 4800 * 81 * 4)
    pe = d["patch_embed"]
    assert pe["parity"]["within_bar"] and pe["fast_mode"]["tokens_equal_fp32_form_rounded_once"] and pe["fast_mode"]["hbm_GB_per_frame"] <= 0.2
    assert d["decoder_loop"]["hip_us_interleaved"]["convex_upsample"] < 25
