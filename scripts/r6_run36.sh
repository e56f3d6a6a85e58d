#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline --no-kernel-events"
run() { # label, steps, env...
  lab=$1; st=$2; shift; shift
  env "$@" timeout 200 python bench.py --steps $st --warmup 5 $Q > gpurun_out/r06w_$lab.json 2> gpurun_out/r06w_$lab.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06w_$lab.json") if l.startswith("{")][-1])
h = d.get("host") or {}
print("$lab", "steps", $st, "value", d["value"], "issue", h.get("host_issue_us_per_frame"), "wait", h.get("host_flow_control_wait_us_per_frame"))
PY
}
for rep in 1 2 3; do
run inline_300_$rep 300 X=1
run async_300_$rep 300 MV_PIPE_ASYNC_BACKEND=1
done
for rep in 1 2 3; do
run inline_20_$rep 20 X=1
run async_20_$rep 20 MV_PIPE_ASYNC_BACKEND=1
done
run async_2core 20 MV_PIPE_ASYNC_BACKEND=1 taskset -c 0,1
run inline_2core 20 taskset -c 0,1
run async_1core 20 MV_PIPE_ASYNC_BACKEND=1 taskset -c 0
