#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline --no-kernel-events"
run() { # label, steps, env...
  lab=$1; st=$2; shift; shift
  env "$@" timeout 200 python bench.py --steps $st --warmup 5 $Q > gpurun_out/r06n_$lab.json 2> gpurun_out/r06n_$lab.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06n_$lab.json") if l.startswith("{")][-1])
h = d.get("host") or {}
print("$lab", "steps", $st, "value", d["value"], "issue", h.get("host_issue_us_per_frame"), "wait", h.get("host_flow_control_wait_us_per_frame"))
PY
}
export MV_PIPE_FRONT_ON=decoder
for rep in 1 2; do
for a in 1 2 3 4; do run fdec_a${a}_$rep 20 MV_PIPE_DD_AHEAD=$a; done
run fdec_async_$rep 20 MV_PIPE_ASYNC_BACKEND=1
done
for a in 1 2 3 4; do run fdec_a${a}_300 300 MV_PIPE_DD_AHEAD=$a; done
run fdec_async_300 300 MV_PIPE_ASYNC_BACKEND=1
run fdec_1core 20 taskset -c 0
run fdec_2core 20 taskset -c 0,1
run fdec_l2_300 300 X=1 
