#!/bin/bash
# Runs on the GPU box (via gpurun): roctx marker + kernel trace of a short bench run with the decoder-loop leg
# (ranges named like the reference's torch.cuda.nvtx.range scopes, covhead.py:90-135) and the frame-stage ranges (MV_TRACE_RANGES=1)
set -u
TAG=${1:-markers}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
( cd /tmp && MV_TRACE_RANGES=1 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT" -o trace -- python "$OLDPWD/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --config4-steps 0 --fast-mode-steps 0 ) > "$OUT/bench.log" 2>&1
ls "$OUT"
f=$(find "$OUT" -name "*marker*stats*.csv" | head -1)
[ -n "$f" ] && cp "$f" "gpurun_out/${TAG}_marker_stats.csv" && cut -c1-140 "$f" | head -16
