#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of bench.py, summaries copied to profiles/ by the caller.
# usage: scripts/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT" -o trace -- python "$OLDPWD/bench.py" --steps 100 --warmup 10 --no-cpu-baseline "$@" ) > "$OUT/bench.log" 2>&1
tail -2 "$OUT/bench.log"
find "$OUT" -name "*kernel_stats.csv" | head -3
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "gpurun_out/${TAG}_kernel_stats.csv" && cut -c1-160 "$f" | head -16
