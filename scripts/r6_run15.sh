#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_device_draw.py -q -x 2>&1 | tail -4
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline --no-kernel-events"
run() { # label, steps, env...
  lab=$1; st=$2; shift; shift
  env "$@" timeout 200 python bench.py --steps $st --warmup 5 $Q > gpurun_out/r06m_$lab.json 2> gpurun_out/r06m_$lab.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06m_$lab.json") if l.startswith("{")][-1])
h = d.get("host") or {}
print("$lab", "steps", $st, "value", d["value"], "issue", h.get("host_issue_us_per_frame"), "wait", h.get("host_flow_control_wait_us_per_frame"))
PY
}
for rep in 1 2 3; do
run fdec_$rep 20 MV_PIPE_FRONT_ON=decoder
run dec_$rep 20 X=1
run dd0_$rep 20 MV_PIPE_DEVICE_DRAW=0
done
for rep in 1 2; do
run fdec_300_$rep 300 MV_PIPE_FRONT_ON=decoder
run dec_300_$rep 300 X=1
run dd0_300_$rep 300 MV_PIPE_DEVICE_DRAW=0
done
