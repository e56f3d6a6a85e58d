#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline --no-kernel-events"
run() { # label, steps, env...
  lab=$1; st=$2; shift; shift
  env "$@" timeout 200 python bench.py --steps $st --warmup 5 $Q > gpurun_out/r06y_$lab.json 2> gpurun_out/r06y_$lab.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06y_$lab.json") if l.startswith("{")][-1])
h = d.get("host") or {}
print("$lab", "steps", $st, "value", d["value"], "issue", h.get("host_issue_us_per_frame"), "wait", h.get("host_flow_control_wait_us_per_frame"))
PY
}
for rep in 1 2; do
run base_$rep 300 X=1
for pace in 2 3 4; do for lag in 3 4 6; do run pace${pace}_lag${lag}_$rep 300 MV_PIPE_GEMM_PACE=$pace MV_PIPE_DD_AHEAD=$lag; done; done
done
