#!/bin/bash
# the three bench lines of the closing evidence by themselves (the driver's 20-step command twice, the default line)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r06}
for i in 1 2; do timeout 500 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_steps20_line_$i.json 2>/dev/null; done
timeout 500 python bench.py > gpurun_out/${T}_bench_default_line.json 2>/dev/null
python - "$T" <<'PY'
import json, sys
T = sys.argv[1]
for f in (f"gpurun_out/{T}_bench_steps20_line_1.json", f"gpurun_out/{T}_bench_steps20_line_2.json", f"gpurun_out/{T}_bench_default_line.json"):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1]); t = d["timed_region"]; e = (d.get("end_to_end") or {}).get("hooked") or {}
    print(d["steps"], d["value"], d["period_us_timed_pass"], t["elapsed_us"], t["gather_issued_us"], "cfg4", (d.get("config4") or {}).get("value"), "e2e", e.get("fps"),
          d["roofline"]["frac"], d["roofline"]["rocprofv3_committed"]["frac_algorithmic"], d["parity"]["within_north_star"], d["cpu_baseline"]["kind"], d["cpu_baseline"]["value"],
          (d.get("plugin_path") or {}).get("ms_per_run_pair_median"))
PY
