#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { lab=$1; shift
  timeout 600 python bench.py --steps 20 --warmup 5 "$@" > gpurun_out/r06t_$lab.json 2> gpurun_out/r06t_$lab.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06t_$lab.json") if l.startswith("{")][-1])
e = d.get("end_to_end") or {}
print("$lab", "value", d["value"], "e2e hooked", (e.get("hooked") or {}).get("ms_per_frame"), "unhooked", (e.get("unhooked") or {}).get("ms_per_frame"), "plugin median", (d.get("plugin_path") or {}).get("ms_per_run_pair_median"), "mean", (d.get("plugin_path") or {}).get("ms_per_run_pair"))
PY
}
run full
run noplugin --plugin-frames 0
run noref --reference-frames 0
run nocfg4dec --config4-steps 0 --no-decoder-leg --exact-steps 0
