#!/bin/bash
# Round-4: patch embed with the chunk-major / parity-split LDS maps, out16 with even / odd column sets.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/r04_pe3.log; : > $L
timeout 600 python -m pytest tests/test_gpu_patch_embed.py tests/test_gpu_fastmode.py -x -q 2>&1 | tail -15 >> $L
timeout 300 python tools/kernel_bench.py volume_f16 patch_embed --iters 30 2>&1 | grep -v amdgpu.ids >> $L
bash scripts/pmc_gpu.sh r04_patch_embed2 patch_embed 2>&1 | grep -A22 "cost_patch_embed" >> $L
for A in "--feat-dtype f16 --layout hwc --volume-store encoder"; do
  timeout 300 python bench.py --height 720 --width 1280 --steps 60 --warmup 10 --no-cpu-baseline --config4-steps 0 --no-decoder-leg --exact-steps 0 $A 2>&1 | tail -1 > gpurun_out/r04_bench_720p_f16hwc_enc16_line.json
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --config4-steps 0 --no-decoder-leg --exact-steps 0 $A 2>&1 | tail -1 > gpurun_out/r04_bench_480p_f16hwc_enc16_line.json
done
python - >> $L 2>&1 <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_bench_*enc16_line.json")):
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("/")[-1], d["value"], "fps", d["ms_per_step"], "ms |", r["kernel"], r["avg_launch_us"], "us frac", r["frac"], "| timeline", d.get("timeline"))
PY
cat $L
