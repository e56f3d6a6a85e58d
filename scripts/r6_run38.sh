#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline"
for i in $(seq 1 16); do MV_BENCH_TRACE=1 timeout 200 python bench.py --steps 20 --warmup 5 $Q 2>&1 >/dev/null | grep "bench trace" | sed 's/.*(us): //' | cut -c1-300; done
