#!/bin/bash
# round 5, closing run after the two-decoder-stream layout became the default: whole GPU suite, smoke, the driver-style 20-step line, kernel trace of the same command
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
timeout 420 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r05c_suite.log; tail -3 gpurun_out/r05c_suite.log
timeout 60 python __graft_entry__.py smoke 2>&1 | tail -1 | tee gpurun_out/r05c_smoke.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05c_bench_steps20_line.json 2> gpurun_out/r05c_bench_steps20.err; echo "steps20 rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05c_bench_steps20_line.json") if l.startswith("{")][-1])
print("steps20 value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "gemm us", d["roofline"]["avg_launch_us"], "north_star", d["parity"]["within_north_star"], "config4", d["config4"]["value"] if d.get("config4") else None)
PY
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline"
OUT=gpurun_out/kprof_r05c; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT" -o trace -- python $R/bench.py --steps 20 --warmup 5 $Q ) > $OUT/run.log 2>&1
cp $(find $OUT -name "*kernel_stats.csv" | head -1) gpurun_out/r05c_bench_kernel_stats.csv 2>/dev/null; grep '^{' $OUT/run.log | tail -1 > gpurun_out/r05c_bench_traced_line.json
rm -rf $OUT
head -8 gpurun_out/r05c_bench_kernel_stats.csv | cut -c1-200
