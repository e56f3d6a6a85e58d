#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline --no-kernel-events"
run() { # label, steps, env...
  lab=$1; st=$2; shift; shift
  env "$@" timeout 200 python bench.py --steps $st --warmup 5 $Q > gpurun_out/r06c_$lab.json 2> gpurun_out/r06c_$lab.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06c_$lab.json") if l.startswith("{")][-1])
print("$lab", "steps", $st, "value", d["value"], "ms/step", d["ms_per_step"])
PY
}
for rep in 1 2 3; do
run dd1_a2_async_$rep 20 MV_PIPE_DEVICE_DRAW=1 MV_PIPE_DD_AHEAD=2
run dd1_a2_inline_$rep 20 MV_PIPE_DEVICE_DRAW=1 MV_PIPE_DD_AHEAD=2 MV_PIPE_ASYNC_BACKEND=0
run dd1_a3_async_$rep 20 MV_PIPE_DEVICE_DRAW=1 MV_PIPE_DD_AHEAD=3
run dd1_a3_inline_$rep 20 MV_PIPE_DEVICE_DRAW=1 MV_PIPE_DD_AHEAD=3 MV_PIPE_ASYNC_BACKEND=0
run dd0_$rep 20 MV_PIPE_DEVICE_DRAW=0
done
run dd1_a2_async_300 300 MV_PIPE_DEVICE_DRAW=1 MV_PIPE_DD_AHEAD=2
run dd1_a2_inline_300 300 MV_PIPE_DEVICE_DRAW=1 MV_PIPE_DD_AHEAD=2 MV_PIPE_ASYNC_BACKEND=0
run dd1_a3_inline_300 300 MV_PIPE_DEVICE_DRAW=1 MV_PIPE_DD_AHEAD=3 MV_PIPE_ASYNC_BACKEND=0
run dd0_300 300 MV_PIPE_DEVICE_DRAW=0
# two host cores only: how much each form depends on the host
run dd1_a2_inline_2cores 20 MV_PIPE_DEVICE_DRAW=1 MV_PIPE_DD_AHEAD=2 MV_PIPE_ASYNC_BACKEND=0 taskset -c 0,1
run dd1_a2_async_2cores 20 MV_PIPE_DEVICE_DRAW=1 MV_PIPE_DD_AHEAD=2 taskset -c 0,1
run dd0_2cores 20 MV_PIPE_DEVICE_DRAW=0 taskset -c 0,1
run dd1_a2_inline_1core 20 MV_PIPE_DEVICE_DRAW=1 MV_PIPE_DD_AHEAD=2 MV_PIPE_ASYNC_BACKEND=0 taskset -c 0
run dd0_1core 20 MV_PIPE_DEVICE_DRAW=0 taskset -c 0
nproc
