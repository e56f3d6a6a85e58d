#!/bin/bash
# round 6, closing evidence: whole GPU suite, smoke, the driver-style 20-step line (x2), the default line, kernel trace of the 20-step command, PMC of the dominant kernel
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
T=${1:-r06}
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/${T}_suite.log; tail -3 gpurun_out/${T}_suite.log
timeout 60 python __graft_entry__.py smoke 2>&1 | tail -1 | tee gpurun_out/${T}_smoke.log
for i in 1 2; do
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_steps20_line_$i.json 2> gpurun_out/${T}_bench_steps20_$i.err; echo "steps20 rc=$?"
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/${T}_bench_steps20_line_$i.json") if l.startswith("{")][-1])
print("steps20 value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"].get("frac_algorithmic"), "gemm us", d["roofline"]["avg_launch_us"], "north_star", d["parity"]["within_north_star"], "config4", d["config4"]["value"] if d.get("config4") else None, "host", {k: v for k, v in (d.get("host") or {}).items() if k != "note"}, "period", d.get("period_us_timed_pass"), "e2e", d.get("end_to_end_fps"))
PY
done
timeout 400 python bench.py > gpurun_out/${T}_bench_default_line.json 2> gpurun_out/${T}_bench_default.err; echo "default rc=$?"
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/${T}_bench_default_line.json") if l.startswith("{")][-1])
print("default steps", d["steps"], "value", d["value"], "ms/step", d["ms_per_step"], "period", d.get("period_us_timed_pass"))
PY
Q="--exact-steps 0 --config4-steps 0 --fast-mode-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline"
OUT=gpurun_out/kprof_$T; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT" -o trace -- python $R/bench.py --steps 20 --warmup 5 $Q ) > $OUT/run.log 2>&1
cp $(find $OUT -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_bench_kernel_stats.csv 2>/dev/null; grep '^{' $OUT/run.log | tail -1 > gpurun_out/${T}_bench_traced_line.json
rm -rf $OUT
head -9 gpurun_out/${T}_bench_kernel_stats.csv | cut -c1-160
bash scripts/pmc_gpu.sh ${T}_split volume_split > gpurun_out/${T}_pmc_split.log 2>&1; cp gpurun_out/pmc_${T}_split.json gpurun_out/${T}_pmc_corr_volume_split_f16x2.json 2>/dev/null; tail -4 gpurun_out/${T}_pmc_split.log | cut -c1-300
rm -rf gpurun_out/pmc_${T}_split_*
