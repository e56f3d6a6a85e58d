#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline --no-kernel-events"
run() { # label, steps, env...
  lab=$1; st=$2; shift; shift
  env "$@" timeout 200 python bench.py --steps $st --warmup 5 $Q > gpurun_out/r06k_$lab.json 2> gpurun_out/r06k_$lab.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06k_$lab.json") if l.startswith("{")][-1])
h = d.get("host") or {}
print("$lab", "steps", $st, "value", d["value"], "issue", h.get("host_issue_us_per_frame"), "wait", h.get("host_flow_control_wait_us_per_frame"))
PY
}
for rep in 1 2 3; do
run dd1_cand_$rep 20 X=1
run dd1_front_$rep 20 MV_PIPE_DD_WAIT_FRONT=1
run dd1_cand_async_$rep 20 MV_PIPE_ASYNC_BACKEND=1
run dd0_$rep 20 MV_PIPE_DEVICE_DRAW=0
done
for rep in 1 2; do
run dd1_cand_300_$rep 300 X=1
run dd1_cand_async_300_$rep 300 MV_PIPE_ASYNC_BACKEND=1
run dd1_cand_a1_300_$rep 300 MV_PIPE_DD_AHEAD=1
run dd0_300_$rep 300 MV_PIPE_DEVICE_DRAW=0
done
