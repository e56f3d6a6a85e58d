#!/bin/bash
# Round-4 evidence run: GPU suite, bench lines (default / 20 steps / configs[2] Fast mode with fp32- and fp16-stored volume), rocprofv3 kernel stats of
# the bench, PMC passes of the split GEMM and of the fused patch embedding.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/r04_suite.log; : > $L
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 >> $L
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3 >> $L
timeout 300 python tools/kernel_bench.py volume_f16 patch_embed --iters 30 2>&1 | grep -v amdgpu.ids >> $L
timeout 600 python bench.py 2>&1 | tail -1 > gpurun_out/r04_bench_default_line.json
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r04_bench_steps20_line.json
for V in "f16hwc_fp32:--feat-dtype f16 --layout hwc" "f16hwc_enc16:--feat-dtype f16 --layout hwc --volume-store encoder"; do
  T=${V%%:*}; A=${V#*:}
  timeout 300 python bench.py --height 720 --width 1280 --steps 60 --warmup 10 --no-cpu-baseline --config4-steps 0 --no-decoder-leg --exact-steps 0 $A 2>&1 | tail -1 > gpurun_out/r04_bench_720p_${T}_line.json
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --config4-steps 0 --no-decoder-leg --exact-steps 0 $A 2>&1 | tail -1 > gpurun_out/r04_bench_480p_${T}_line.json
done
bash scripts/profile_gpu.sh r04_bench --config4-steps 0 --no-decoder-leg --exact-steps 0 >> $L 2>&1
grep '^{"metric' gpurun_out/prof_r04_bench/bench.log | tail -1 > gpurun_out/r04_bench_profiled_line.json
MV_SPLIT_MODE=f16x2 bash scripts/pmc_gpu.sh r04_split_f16x2 volume_split >> $L 2>&1
bash scripts/pmc_gpu.sh r04_patch_embed patch_embed >> $L 2>&1
bash scripts/profile_kernels_gpu.sh r04_kernels volume_split volume_f16 patch_embed lookup pgo >> $L 2>&1
python - >> $L 2>&1 <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_bench_*_line.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d["roofline"]
    print(f.split("/")[-1], d["value"], "fps", d["ms_per_step"], "ms |", r["kernel"], r["avg_launch_us"], "us frac", r["frac"], "alone", r.get("isolated_avg_launch_us"), "| timeline", d.get("timeline"))
    if d.get("parity"):
        p = d["parity"]; print("    parity kp", p["keypoints_bit_exact_frames"], "/", p["frames"], p["max_pose_dt_m"], "| vol/lookup", p.get("volume_and_lookups"), "| vs ref", {k: v for k, v in (p.get("vs_reference_loop") or {}).items() if k != "what"})
        print("    cpu_baseline", d["cpu_baseline"])
PY
cat $L
