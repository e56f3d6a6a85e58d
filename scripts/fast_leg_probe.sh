#!/bin/bash
# probe: the bench line's fast_mode leg in the line's order (in front of configs[4]); MV_FAST_PROBE_HWQ=1 also runs it with more HIP hardware queues allowed (one-off, round 6)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
Q="--exact-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline --no-kernel-events"
for spec in "line_order:--config4-steps 60:X=1" "line_order_hwq8:--config4-steps 60:GPU_MAX_HW_QUEUES=8"; do
  lab=${spec%%:*}; rest=${spec#*:}; fl=${rest%%:*}; ev=${rest#*:}
  env $ev python bench.py --steps 20 --warmup 5 $Q $fl --fast-mode-steps 200 > gpurun_out/fl_$lab.json 2> gpurun_out/fl_$lab.err
  python - "$lab" <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/fl_{sys.argv[1]}.json") if l.startswith("{")][-1])
print(sys.argv[1], d["value"], "config4", d["config4"]["value"], json.dumps({k: v for k, v in d["fast_mode"].items() if k in ("one_lane", "lanes_32", "error")}))
PY
done
