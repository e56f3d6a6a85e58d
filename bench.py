#!/usr/bin/env python
"""bench.py — stereo frames/sec of the MAC-VO hot path on MI355X (contract: see the task statement / DESIGN.md §Measurement).

One "step" = one ``run_pair`` of the hot path on one 640x480 stereo frame (BASELINE.json configs[1]):
  2 all-pairs cost volumes (stereo + temporal pair, B = 2 in one launch) + 12 x 9x9 window lookups + frontend
  epilogue + covariance-aware keypoint selection (200 pts, host randperm) + tracking gathers + 2 x covariance model +
  observation filter + covariance-weighted two-frame PGO (LM, <= 10 steps), all in hand-written HIP kernels behind
  the C ABI, inputs (feature maps, lookup coordinates, network flow / log-sigma) already resident in HBM.
The learned FlowFormer layers are not part of the step (their source and weights are absent from the reference).

N > 1: one process per GPU (torch.distributed / RCCL), each rank owns an independent sequence (weak scaling);
the only collective is one all_gather of the per-frame poses at the end of the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

# MI355X peaks (/opt/skills/guides/MI355X_MICROARCH.md §Chip-level parameters)
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--feat-dtype", choices=["f32", "f16", "bf16"], default="f32")
    ap.add_argument("--layout", choices=["chw", "hwc"], default="chw")
    ap.add_argument("--volume-precision", choices=["exact", "split3", "split2"], default="exact",
                    help="fp32 features: exact fp32 MFMA (default) or bf16x3 split (fp32-class accuracy; needs --layout hwc)")
    ap.add_argument("--graph", choices=["disp", "reproj", "icp"], default="disp")
    ap.add_argument("--pool", type=int, default=24, help="distinct synthetic frames (closed trajectory) kept in HBM")
    ap.add_argument("--cpu-frames", type=int, default=120, help="frames timed for the CPU baseline (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graphs", action="store_true", help="replay the decoder-side segment (12 lookups + epilogue + selector) as a hipGraph (measured slower than eager launches on ROCm 7.2: 2.46 k vs 2.60 k fps)")
    ap.add_argument("--driver", choices=["native", "python"], default="native",
                    help="host-side frame sequencing: the C++ driver (mv_frame_pipe_*) or the Python loop over the per-op entry points")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip per-launch HIP events around the volume kernel")
    return ap.parse_args()


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP hot path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run (also exercises RCCL at world = 1)
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group(backend="nccl", device_id=dev)
        dist = dist_mod

    from macvo_amd import ops
    from macvo_amd.distributed import gather_poses
    from macvo_amd.pipeline import Camera, FrameInputs, HotPath, HotPathConfig, NativeHotPath
    from tests import synth

    H, W, C = args.height, args.width, args.channels
    fdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[args.feat_dtype]
    # ---- synthetic, seeded, per-rank sequence (closed trajectory so the pool can be cycled without a seam)
    cam, frames_cpu, _ = synth.make_sequence(args.pool, H, W, C=C, iters=args.iters, seed=1000 + rank, feat_dtype=fdt,
                                             pool=2, closed_loop=True)
    if args.layout == "hwc":
        for fr in frames_cpu:
            fr["fmap1"] = fr["fmap1"].permute(0, 2, 3, 1).contiguous()
            fr["fmap2"] = fr["fmap2"].permute(0, 2, 3, 1).contiguous()
    cache = {}

    def to_dev(t):
        k = t.data_ptr()
        if k not in cache:
            cache[k] = t.to(dev)
        return cache[k]

    use_graphs = args.graphs
    assert args.pool % 6 == 0 or not use_graphs, "--pool must be a multiple of 6 with graphs (one graph per resident frame)"
    frames = [FrameInputs(static=True, **{k: to_dev(v) for k, v in fr.items()}) for fr in frames_cpu]
    native = args.driver == "native"
    assert not (native and use_graphs), "--graphs belongs to the Python driver"
    hot = (NativeHotPath if native else HotPath)(
        Camera(**cam), HotPathConfig(graph_type=args.graph, feature_layout=args.layout,
                                     volume_precision=args.volume_precision, use_graphs=use_graphs), dev)
    torch.manual_seed(1234 + rank)  # the selector consumes the global CPU generator (reference behaviour)

    # ---- per-launch HIP events around the dominant kernel (cost volume), on the launch stream
    vol_events = []
    orig_corr_volume = ops.corr_volume
    record = {"on": False}

    def timed_corr_volume(*a, **k):
        if not record["on"]:
            return orig_corr_volume(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig_corr_volume(*a, **k)
        e1.record()
        vol_events.append((e0, e1))
        return out

    if not args.no_kernel_events and not native:
        ops.corr_volume = timed_corr_volume

    hot.initialize(frames[0])
    t_idx = 1
    if use_graphs:
        # setup (untimed, before the warm-up): one pass over the resident frames captures their decoder-side hipGraphs
        for _ in hot.run(frames[(t_idx + k) % args.pool] for k in range(args.pool)):
            pass
        t_idx += args.pool
    poses = torch.zeros((args.steps, 7), dtype=torch.float32, device=dev)
    for _ in hot.run(frames[(t_idx + k) % args.pool] for k in range(args.warmup)):
        pass
    t_idx += args.warmup

    def barrier():
        if dist is not None:
            dist.barrier()

    if dist is not None:  # warm the collectives used in / around the timed region (RCCL sets channels up lazily)
        gather_poses(torch.zeros((args.steps, 7), dtype=torch.float32, device=dev), dist)
        dist.all_reduce(torch.zeros(1, dtype=torch.float64, device=dev), op=dist.ReduceOp.MAX)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    record["on"] = True
    if native and not args.no_kernel_events:
        hot.time_volume(args.steps)   # HIP-event pairs around the volume GEMM, recorded by the driver on its GEMM stream
    t0 = time.perf_counter()
    # software-pipelined stream (frame t+1's frontend is queued before frame t's host randperm); K full run_pairs
    for _ in hot.run((frames[(t_idx + k) % args.pool] for k in range(args.steps)), pose_sink=poses):
        pass
    all_poses, _ = gather_poses(poses, dist)  # the one collective of the job (no-op for N = 1)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    record["on"] = False
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert torch.isfinite(all_poses).all(), "non-finite pose in the benchmark stream"

    # ---- roofline of the dominant kernel (cost volume): algorithmic work per launch / measured launch duration
    h8, w8 = H // 8, W // 8
    n_q = h8 * w8
    pairs = 2
    flops_per_launch = pairs * 2.0 * n_q * n_q * C                       # 2*N^2*C per pair
    esz = 4 if args.feat_dtype == "f32" else 2
    bytes_per_launch = pairs * (2.0 * n_q * C * esz + 4.0 * n_q * n_q)   # read f1,f2 + write fp32 volume
    # HBM traffic of the dominant kernel from a committed rocprofv3 --pmc pass over the same launch configuration
    # (scripts/pmc_gpu.sh -> profiles/r01_pmc_corr_volume.json; PMC passes cannot be mixed into this run)
    traffic = None
    try:
        if H == 480 and W == 640 and C == 256 and args.feat_dtype == "f32" and args.volume_precision == "exact":
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_corr_volume.json")))
            traffic = pm["corr_volume_f32_" + args.layout]["_derived"]["traffic_bytes_per_launch"]
    except Exception:  # noqa: BLE001
        traffic = None
    roofline = None
    if native and not args.no_kernel_events:
        vol_events = hot.volume_times_ms()
    if vol_events:
        ms = vol_events if native else [a.elapsed_time(b) for a, b in vol_events]
        avg_s = sum(ms) / len(ms) / 1e3
        if args.feat_dtype == "f32" and args.volume_precision in ("split3", "split2"):
            ach = flops_per_launch / avg_s / 1e12
            nprod = 6.0 if args.volume_precision == "split3" else 3.0
            eff_peak = 2500.0 / nprod   # bf16 MFMA products executed per algorithmic product at the 2.5 PFLOP/s dense bf16 peak
            roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": round(eff_peak, 1), "unit": "TFLOP/s",
                        "frac": round(ach / eff_peak, 4), "traffic": None,
                        "kernel": f"{args.volume_precision} pre-pass + corr_volume_bf16x3_hwc<{int(nprod) // 3 + 1}>",
                        "avg_launch_us": round(avg_s * 1e6, 2), "launches": len(ms),
                        "algorithmic_flops_per_launch": flops_per_launch, "algorithmic_bytes_per_launch": bytes_per_launch,
                        "note": "achieved = algorithmic fp32 FLOPs / time; peak = 2500 TFLOP/s bf16 dense / executed products per algorithmic product"}
        elif args.feat_dtype == "f32":
            ach = flops_per_launch / avg_s / 1e12
            roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                        "kernel": "corr_volume_f32_" + args.layout, "avg_launch_us": round(avg_s * 1e6, 2),
                        "launches": len(ms), "algorithmic_flops_per_launch": flops_per_launch,
                        "algorithmic_bytes_per_launch": bytes_per_launch}
        else:
            ach = bytes_per_launch / avg_s / 1e9
            roofline = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None,
                        "kernel": "corr_volume_h_" + args.layout, "avg_launch_us": round(avg_s * 1e6, 2),
                        "launches": len(ms), "algorithmic_flops_per_launch": flops_per_launch,
                        "algorithmic_bytes_per_launch": bytes_per_launch}

    # ---- CPU baseline: the oracle pipeline (torch-CPU ops shaped like the reference) on a bounded sample
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.pipeline import OracleHotPath

        # 8 threads = what the reference's optimizer child uses (Optimization/Interface.py:254); more threads on a
        # many-core host make the small float64 LM ops slower (fork/join dominated), measured 34 s/frame at 256 threads
        cores = min(os.cpu_count() or 1, 8)
        torch.set_num_threads(cores)
        ora = OracleHotPath(cam, dict(graph_type=args.graph))
        cfr = [{k: (v.float() if v.dtype in (torch.float16, torch.bfloat16) else v) for k, v in fr.items()} for fr in frames_cpu]
        if args.layout == "hwc":
            for fr in cfr:
                fr["fmap1"] = fr["fmap1"].permute(0, 3, 1, 2).contiguous()
                fr["fmap2"] = fr["fmap2"].permute(0, 3, 1, 2).contiguous()
        torch.manual_seed(1234)
        ora.initialize(cfr[0])
        ora.step(cfr[1])  # warm-up (thread pools, first-call overheads)
        c0 = time.perf_counter()
        ncpu = 0
        while ncpu < args.cpu_frames and (time.perf_counter() - c0) < 25.0:
            ora.step(cfr[(2 + ncpu) % args.pool])
            ncpu += 1
        csec = time.perf_counter() - c0
        cpu_baseline = {"value": round(ncpu / csec, 3), "unit": "stereo frames/s", "cores": cores, "kind": "port",
                        "sample": f"{ncpu} frames of the same {W}x{H} workload through oracle/pipeline.py "
                                  f"(torch-CPU einsum volume, grid_sample lookup, max_pool2d selector, float64 dense-weight LM), "
                                  f"{csec:.1f} s"}

    if rank == 0:
        total_frames = world * args.steps
        line = {
            "metric": "stereo frames/sec at 640x480 (hot path: cost volume + lookup, keypoint selection, covariance, PGO)",
            "value": round(total_frames / elapsed, 2),
            "unit": "stereo frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f32": "f32 (volume/lookup/covariance) + f64 (PGO)", "f16": "f16 in / f32 acc (volume) + f32 + f64 (PGO)",
                      "bf16": "bf16 in / f32 acc (volume) + f32 + f64 (PGO)"}[args.feat_dtype],
            "data": "synthetic (seeded planar-scene stereo stream, random feature maps; no weights/datasets available)",
            "config": {"workload": f"configs[1]: single MI355X, {W}x{H} synthetic stereo, HIP correlation volume + GN backend, 1-seq stream per GPU",
                       "per_step": f"2 cost volumes [{n_q}x{C}x{n_q}] + {args.iters}x2 9x9 lookups + epilogue + CovAwareSelector_NoDepth(200) + 2x MatchCovariance(31x31) + TwoFrame_PGO({args.graph})",
                       "feature_dtype": args.feat_dtype, "feature_layout": args.layout, "volume_precision": args.volume_precision,
                       "hip_graphs": use_graphs, "host_driver": args.driver,
                       "excluded": "learned FlowFormer layers (source + weights absent from the reference checkout)",
                       "parallelism": f"{world} independent sequence(s), one per GPU; one all_gather of poses"},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
