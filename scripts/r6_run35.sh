#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_device_draw.py tests/test_gpu_native.py tests/test_gpu_lanes.py -q -x 2>&1 | tail -4
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline --no-kernel-events"
run() { # label, steps, env...
  lab=$1; st=$2; shift; shift
  env "$@" timeout 200 python bench.py --steps $st --warmup 5 $Q > gpurun_out/r06v_$lab.json 2> gpurun_out/r06v_$lab.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06v_$lab.json") if l.startswith("{")][-1])
h = d.get("host") or {}
print("$lab", "steps", $st, "value", d["value"], "issue", h.get("host_issue_us_per_frame"), "wait", h.get("host_flow_control_wait_us_per_frame"))
PY
}
for rep in 1 2 3; do
run lean_spin_$rep 20 X=1
run lean_nospin_$rep 20 MV_PIPE_FLOW_SPIN_US=0
run nolean_spin_$rep 20 MV_PIPE_LEAN=0
run nolean_nospin_$rep 20 MV_PIPE_LEAN=0 MV_PIPE_FLOW_SPIN_US=0
done
for rep in 1 2; do
run lean_spin_300_$rep 300 X=1
run lean_nospin_300_$rep 300 MV_PIPE_FLOW_SPIN_US=0
run nolean_spin_300_$rep 300 MV_PIPE_LEAN=0
run nolean_nospin_300_$rep 300 MV_PIPE_LEAN=0 MV_PIPE_FLOW_SPIN_US=0
run lean_spin_a1_300_$rep 300 MV_PIPE_DD_AHEAD=1
run lean_spin_a3_300_$rep 300 MV_PIPE_DD_AHEAD=3
done
