#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats of tools/kernel_bench.py at a given configuration
# usage: scripts/profile_kernels_gpu.sh <tag> <kernel_bench args...>
set -u
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/kprof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT" -o trace -- python "$OLDPWD/tools/kernel_bench.py" "$@" ) > "$OUT/bench.log" 2>&1
grep -v "^W2\|rocprofv3\|amdgpu.ids" "$OUT/bench.log" | tail -12
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "gpurun_out/${TAG}_kernel_stats.csv"
cp "$OUT/bench.log" "gpurun_out/${TAG}_kernel_bench.log"
rm -rf "$OUT"
