#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline --no-kernel-events"
run() { # label, steps, env...
  lab=$1; st=$2; shift; shift
  env "$@" timeout 200 python bench.py --steps $st --warmup 5 $Q > gpurun_out/r06l_$lab.json 2> gpurun_out/r06l_$lab.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06l_$lab.json") if l.startswith("{")][-1])
print("$lab", "steps", $st, "value", d["value"])
PY
}
for rep in 1 2 3; do
run late_$rep 20 MV_PIPE_DD_ORDER=late
run late_async_$rep 20 MV_PIPE_DD_ORDER=late MV_PIPE_ASYNC_BACKEND=1
run early_front_$rep 20 MV_PIPE_DD_WAIT_FRONT=1
run dd0_$rep 20 MV_PIPE_DEVICE_DRAW=0
done
for rep in 1 2; do
run late_300_$rep 300 MV_PIPE_DD_ORDER=late
run late_async_300_$rep 300 MV_PIPE_DD_ORDER=late MV_PIPE_ASYNC_BACKEND=1
run dd0_300_$rep 300 MV_PIPE_DEVICE_DRAW=0
done
run late_1core 20 MV_PIPE_DD_ORDER=late taskset -c 0
run early_1core 20 MV_PIPE_DD_WAIT_FRONT=1 taskset -c 0
run late_2core 20 MV_PIPE_DD_ORDER=late taskset -c 0,1
