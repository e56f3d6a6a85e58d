#!/bin/bash
# usage: prof_probe.sh <tag> <python script> ; kernel-trace stats of a scratch probe (GPU-side durations, no host floor)
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT" -o trace -- python "$R/$1" ) > "$OUT/log.txt" 2>&1
grep -v amdgpu.ids "$OUT/log.txt" | tail -8
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "gpurun_out/${TAG}_kernel_stats.csv" && cut -d, -f1-8 "$f" | cut -c1-200 | head -12
