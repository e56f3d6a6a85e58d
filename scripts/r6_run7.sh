#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
run() { # label, extra
  lab=$1; shift
  timeout 400 python bench.py --steps 20 --warmup 5 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 "$@" > gpurun_out/r06g_$lab.json 2> gpurun_out/r06g_$lab.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06g_$lab.json") if l.startswith("{")][-1])
c = d.get("config4") or {}
print("$lab", "value", d["value"], "config4", c.get("value"), (c.get("timeline") or {}).get("period_us"))
PY
}
run A
run B --exact-steps 0
run C --exact-steps 0 --no-cpu-baseline
run D --exact-steps 0 --no-cpu-baseline --parity-frames 0
