#!/bin/bash
# 32-lane pipe (configs[4]) with the per-frame timeline, row-major vs tiled volume (probe, round 6)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
Q="--exact-steps 0 --config4-steps 0 --fast-mode-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline --parity-frames 0"
for t in 0 1; do
  MV_PIPE_TILED=$t python bench.py --lanes 32 --steps 40 --warmup 5 $Q ${C4_EXTRA:-} > gpurun_out/c4tl_$t.json 2> gpurun_out/c4tl_$t.err
  python - $t <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/c4tl_{sys.argv[1]}.json") if l.startswith("{")][-1])
print("tiled", sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "gemm us", d["roofline"].get("avg_launch_us"), "timeline", json.dumps(d.get("timeline")))
PY
done
