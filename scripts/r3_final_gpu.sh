#!/bin/bash
# Round-3 evidence run on the GPU box: full GPU test suite, the bench lines, rocprofv3 kernel stats and PMC passes.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/r03_final.log; : > $L
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 >> $L
timeout 600 python bench.py 2>&1 | tail -1 > gpurun_out/r03_bench_default_line.json
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r03_bench_steps20_line.json
bash scripts/profile_gpu.sh r03_bench --config4-steps 0 --no-decoder-leg --exact-steps 0 >> $L 2>&1
tail -1 gpurun_out/prof_r03_bench/bench.log > gpurun_out/r03_bench_profiled_line.json
MV_SPLIT_MODE=f16x2 bash scripts/pmc_gpu.sh r03_split_f16x2 volume_split >> $L 2>&1
MV_SPLIT_MODE=bf16x3 bash scripts/pmc_gpu.sh r03_split_bf16x3 volume_split >> $L 2>&1
bash scripts/profile_kernels_gpu.sh r03_kernels volume_split volume_f32 lookup pgo >> $L 2>&1
timeout 300 python tools/split_probe.py 2x60x80 >> $L 2>&1
python - >> $L 2>&1 <<'PY'
import json
for f in ("r03_bench_default_line.json", "r03_bench_steps20_line.json"):
    d = json.load(open("gpurun_out/" + f))
    r = d["roofline"]
    print(f, d["value"], "fps", d["ms_per_step"], "ms |", r["kernel"], r["avg_launch_us"], "us frac", r["frac"], "alone", r.get("isolated_avg_launch_us"), "| timeline", d.get("timeline"))
    for k, v in (d.get("other_precisions") or {}).items():
        print("   ", k, v["value"], v["ms_per_step"], v["roofline"]["avg_launch_us"], v["roofline"]["frac"], v["roofline"].get("isolated_avg_launch_us"))
    if d.get("config4"):
        c = d["config4"]; print("    config4", c["value"], c["ms_per_step"], c["roofline"]["avg_launch_us"], c["roofline"]["frac"], c.get("timeline"))
    if d.get("parity"): print("    parity", d["parity"]["keypoints_bit_exact_frames"], d["parity"]["max_pose_dt_m"], d["rte_vs_oracle"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
cat $L
