#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline --no-kernel-events"
run() { # label, steps, env...
  lab=$1; st=$2; shift; shift
  env "$@" timeout 200 python bench.py --steps $st --warmup 5 $Q > gpurun_out/r06j_$lab.json 2> gpurun_out/r06j_$lab.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06j_$lab.json") if l.startswith("{")][-1])
h = d.get("host") or {}
print("$lab", "steps", $st, "value", d["value"], "issue", h.get("host_issue_us_per_frame"), "wait", h.get("host_flow_control_wait_us_per_frame"))
PY
  grep -h "host stats" gpurun_out/r06j_$lab.err | tail -2 | cut -c1-400
}
run dd1_stats 300 MV_PIPE_HOST_STATS=1
for rep in 1 2 3; do
run dd1_$rep 20 X=1
run dd0_$rep 20 MV_PIPE_DEVICE_DRAW=0
done
run dd1_300 300 X=1
run dd0_300 300 MV_PIPE_DEVICE_DRAW=0
run dd1_300b 300 X=1
run dd0_300b 300 MV_PIPE_DEVICE_DRAW=0
