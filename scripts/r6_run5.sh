#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline --no-kernel-events --parity-frames 0"
run() { # label, lanes, steps, env...
  lab=$1; ln=$2; st=$3; shift; shift; shift
  env "$@" timeout 300 python bench.py --lanes $ln --steps $st --warmup 5 $Q > gpurun_out/r06e_$lab.json 2> gpurun_out/r06e_$lab.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06e_$lab.json") if l.startswith("{")][-1])
print("$lab", "lanes", $ln, "steps", $st, "value", d["value"], "ms/step", d["ms_per_step"], d.get("host"))
PY
}
run l32_dd0 32 20 MV_PIPE_DEVICE_DRAW=0
run l32_dd1_a2 32 20 MV_PIPE_DEVICE_DRAW=1 MV_PIPE_DD_AHEAD=2
run l32_dd1_a3 32 20 MV_PIPE_DEVICE_DRAW=1 MV_PIPE_DD_AHEAD=3
run l32_dd1_a4 32 20 MV_PIPE_DEVICE_DRAW=1 MV_PIPE_DD_AHEAD=4
run l32_dd1_a6 32 20 MV_PIPE_DEVICE_DRAW=1 MV_PIPE_DD_AHEAD=6
run l32_dd1_a3_async 32 20 MV_PIPE_DEVICE_DRAW=1 MV_PIPE_DD_AHEAD=3 MV_PIPE_ASYNC_BACKEND=1
run l32_dd1_inf 32 20 MV_PIPE_DEVICE_DRAW=1 MV_PIPE_DD_AHEAD=-1
run l4_dd0 4 40 MV_PIPE_DEVICE_DRAW=0
run l4_dd1_a2 4 40 MV_PIPE_DEVICE_DRAW=1 MV_PIPE_DD_AHEAD=2
run l4_dd1_a3 4 40 MV_PIPE_DEVICE_DRAW=1 MV_PIPE_DD_AHEAD=3
run l2_dd0 2 60 MV_PIPE_DEVICE_DRAW=0
run l2_dd1_a2 2 60 MV_PIPE_DEVICE_DRAW=1 MV_PIPE_DD_AHEAD=2
run l2_dd1_a3 2 60 MV_PIPE_DEVICE_DRAW=1 MV_PIPE_DD_AHEAD=3
