#!/bin/bash
# Runs on the GPU box (via gpurun): per-kernel start/end timeline of a short bench run -> gpurun_out/<tag>_kernel_trace.csv
# usage: scripts/trace_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-t1}; shift || true
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/trace_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
( cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OLDPWD/$OUT" -o trace -- python "$OLDPWD/bench.py" --steps 60 --warmup 10 --no-cpu-baseline "$@" ) > "$OUT/bench.log" 2>&1
tail -1 "$OUT/bench.log" | cut -c1-260
f=$(find "$OUT" -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && cp "$f" "gpurun_out/${TAG}_kernel_trace.csv"
f=$(find "$OUT" -name "*memory_copy_trace.csv" | head -1)
[ -n "$f" ] && cp "$f" "gpurun_out/${TAG}_memcopy_trace.csv"
rm -rf "$OUT"
ls -la gpurun_out/${TAG}_*
