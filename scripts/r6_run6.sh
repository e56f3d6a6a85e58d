#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline --parity-frames 0"
run() { # label, lanes, steps, extra, env...
  lab=$1; ln=$2; st=$3; ex=$4; shift; shift; shift; shift
  env "$@" timeout 300 python bench.py --lanes $ln --steps $st --warmup 10 $Q $ex > gpurun_out/r06f_$lab.json 2> gpurun_out/r06f_$lab.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06f_$lab.json") if l.startswith("{")][-1])
h = d.get("host") or {}
print("$lab", "lanes", $ln, "steps", $st, "value", d["value"], "ms/step", d["ms_per_step"], "issue", h.get("host_issue_us_per_frame"), "wait", h.get("host_flow_control_wait_us_per_frame"), (d.get("timeline") or {}).get("period_us"))
PY
}
run l32_100_ev_dd1 32 100 "" MV_PIPE_DEVICE_DRAW=1
run l32_100_noev_dd1 32 100 "--no-kernel-events" MV_PIPE_DEVICE_DRAW=1
run l32_100_ev_dd0 32 100 "" MV_PIPE_DEVICE_DRAW=0
run l32_100_noev_dd0 32 100 "--no-kernel-events" MV_PIPE_DEVICE_DRAW=0
run l32_100_ev_dd1_a3 32 100 "" MV_PIPE_DEVICE_DRAW=1 MV_PIPE_DD_AHEAD=3
run l32_100_ev_dd1_async 32 100 "" MV_PIPE_DEVICE_DRAW=1 MV_PIPE_ASYNC_BACKEND=1
