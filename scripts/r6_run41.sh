#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline"
OUT=gpurun_out/kprof_r06c; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT" -o trace -- python $R/bench.py --steps 20 --warmup 5 $Q ) > $OUT/run.log 2>&1
cp $(find $OUT -name "*kernel_stats.csv" | head -1) gpurun_out/r06c_bench_kernel_stats.csv; grep '^{' $OUT/run.log | tail -1 > gpurun_out/r06c_bench_traced_line.json
rm -rf $OUT
head -8 gpurun_out/r06c_bench_kernel_stats.csv | cut -c1-60,150-260
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/r06c_s20_$i.json 2>/dev/null; python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06c_s20_$i.json") if l.startswith("{")][-1])
print("steps20", d["value"], d["roofline"]["avg_launch_us"], d["period_us_timed_pass"])
PY
done
