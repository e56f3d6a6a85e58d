#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline --no-kernel-events"
run() { # label, env...
  lab=$1; shift
  env "$@" timeout 200 python bench.py --steps 300 --warmup 5 $Q > gpurun_out/r06b_$lab.json 2> gpurun_out/r06b_$lab.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06b_$lab.json") if l.startswith("{")][-1])
print("$lab", "value", d["value"], "ms/step", d["ms_per_step"])
PY
  grep -h "host stats\|bench trace" gpurun_out/r06b_$lab.err | cut -c1-600
}
run dd1_trace MV_PIPE_DEVICE_DRAW=1 MV_BENCH_TRACE=1 MV_PIPE_HOST_STATS=1
run dd0_trace MV_PIPE_DEVICE_DRAW=0 MV_BENCH_TRACE=1 MV_PIPE_HOST_STATS=1
for a in 0 1 2 3 4 6; do run dd1_ahead$a MV_PIPE_DEVICE_DRAW=1 MV_PIPE_DD_AHEAD=$a; done
run dd1_inline MV_PIPE_DEVICE_DRAW=1 MV_PIPE_ASYNC_BACKEND=0
run dd1_inline_ahead2 MV_PIPE_DEVICE_DRAW=1 MV_PIPE_ASYNC_BACKEND=0 MV_PIPE_DD_AHEAD=2
run dd1_classic MV_PIPE_DEVICE_DRAW=1 MV_PIPE_LAYOUT=classic
