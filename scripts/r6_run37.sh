#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline"
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
timeout 200 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/r06x_$i.json 2> gpurun_out/r06x_$i.err
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06x_$i.json") if l.startswith("{")][-1])
h = d.get("host") or {}
print("run $i value", d["value"], "period", d.get("period_us_timed_pass"), "threads", h.get("host_threads"), "issue", h.get("host_issue_us_per_frame"), "wait", h.get("host_flow_control_wait_us_per_frame"))
PY
done
