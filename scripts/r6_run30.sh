#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { lab=$1; shift
  timeout 500 python bench.py --steps 20 --warmup 5 "$@" > gpurun_out/r06s_$lab.json 2> gpurun_out/r06s_$lab.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06s_$lab.json") if l.startswith("{")][-1])
e = d.get("end_to_end") or {}
print("$lab", "value", d["value"], "e2e hooked", (e.get("hooked") or {}).get("ms_per_frame"), "unhooked", (e.get("unhooked") or {}).get("ms_per_frame"), "plugin", (d.get("plugin_path") or {}).get("ms_per_run_pair_median"))
PY
}
run e2e_only --no-decoder-leg --config4-steps 0 --exact-steps 0 --plugin-frames 0 --cpu-frames 1 --parity-frames 2 --reference-frames 0
run e2e_cfg4 --no-decoder-leg --exact-steps 0 --plugin-frames 0 --cpu-frames 1 --parity-frames 2 --reference-frames 0
run e2e_kern --config4-steps 0 --exact-steps 0 --plugin-frames 0 --cpu-frames 1 --parity-frames 2 --reference-frames 0
