#!/bin/bash
# BASELINE configs[2] (1280x720) through the frame driver: f16x2 default, exact, Fast-mode 16-bit features
cd $GRAFT_REPO_ROOT
for V in "f16x2:--volume-precision f16x2" "exact:--volume-precision exact" "f16hwc:--feat-dtype f16 --layout hwc"; do
  T=${V%%:*}; A=${V#*:}
  timeout 300 python bench.py --height 720 --width 1280 --steps 60 --warmup 10 --no-cpu-baseline --config4-steps 0 --no-decoder-leg --exact-steps 0 $A 2>&1 | tail -1 > gpurun_out/r03_bench_720p_${T}_line.json
  python - $T <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r03_bench_720p_{sys.argv[1]}_line.json")); r=d['roofline']
print(sys.argv[1], d['value'],'fps',d['ms_per_step'],'ms |',r['kernel'],r['avg_launch_us'],'us frac',r['frac'],'alone',r.get('isolated_avg_launch_us'),r.get('isolated_frac'))
PY
done
