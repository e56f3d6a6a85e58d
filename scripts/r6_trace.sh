#!/bin/bash
# kernel trace of the driver's 20-step command (what roofline.rocprofv3_committed reads): profiles/<tag>_bench_kernel_stats.csv + the line printed under the tracer
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
T=${1:-r06}
Q="--exact-steps 0 --config4-steps 0 --fast-mode-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline"
OUT=gpurun_out/kprof_$T; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT" -o trace -- python $R/bench.py --steps 20 --warmup 5 $Q ) > $OUT/run.log 2>&1
cp $(find $OUT -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_bench_kernel_stats.csv 2>/dev/null; grep '^{' $OUT/run.log | tail -1 > gpurun_out/${T}_bench_traced_line.json
rm -rf $OUT
head -12 gpurun_out/${T}_bench_kernel_stats.csv | cut -c1-170
