#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
PROBE_MODE=extras timeout 300 python profiles/probes/r6_queue_history.py 0 1 2>&1 | grep "lane pipe" | tail -1
PROBE_CLOSE=1 PROBE_MODE=extras timeout 300 python profiles/probes/r6_queue_history.py 0 1 2>&1 | grep "lane pipe" | tail -1
MV_PIPE_DEVICE_DRAW=0 PROBE_MODE=extras timeout 300 python profiles/probes/r6_queue_history.py 0 1 2>&1 | grep "lane pipe" | tail -1
timeout 400 python bench.py --steps 20 --warmup 5 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --exact-steps 0 > gpurun_out/r06i.json 2> gpurun_out/r06i.err
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06i.json") if l.startswith("{")][-1])
c = d.get("config4") or {}
print("value", d["value"], "config4", c.get("value"), (c.get("timeline") or {}).get("period_us"))
PY
