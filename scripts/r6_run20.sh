#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline"
for i in 1 2 3 4 5 6 7 8; do
timeout 200 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/r06q_$i.json 2> gpurun_out/r06q_$i.err
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06q_$i.json") if l.startswith("{")][-1])
h = d.get("host") or {}
print("run $i value", d["value"], "period", d.get("period_us_timed_pass"), "issue", h.get("host_issue_us_per_frame"), "gemm", d["roofline"]["avg_launch_us"], d["roofline"].get("median_launch_us"))
PY
done
