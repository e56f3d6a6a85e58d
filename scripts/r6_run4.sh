#!/bin/bash
# round 6: full GPU suite after the device-driven frame + ADVICE fixes; default 20-step line x2; kernel trace
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r06e_suite.log; tail -5 gpurun_out/r06e_suite.log
timeout 60 python __graft_entry__.py smoke 2>&1 | tail -1 | tee gpurun_out/r06e_smoke.log
for i in 1 2; do
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r06e_bench_steps20_$i.json 2> gpurun_out/r06e_bench_steps20_$i.err; echo "steps20 rc=$?"
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06e_bench_steps20_$i.json") if l.startswith("{")][-1])
print("steps20 value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"].get("frac_algorithmic"), "gemm us", d["roofline"]["avg_launch_us"], "north_star", d["parity"]["within_north_star"], "config4", d["config4"]["value"] if d.get("config4") else None, "host", d.get("host"), "period", d.get("period_us_timed_pass"), "e2e", d.get("end_to_end_fps"))
PY
done
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline"
OUT=gpurun_out/kprof_r06e; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT" -o trace -- python $R/bench.py --steps 300 --warmup 5 $Q ) > $OUT/run.log 2>&1
cp $(find $OUT -name "*kernel_stats.csv" | head -1) gpurun_out/r06e_bench300_kernel_stats.csv 2>/dev/null; grep '^{' $OUT/run.log | tail -1 > gpurun_out/r06e_bench300_traced_line.json
rm -rf $OUT
head -14 gpurun_out/r06e_bench300_kernel_stats.csv | cut -c1-180
