#!/bin/bash
# Round-4 A/B: frames in flight (MV_PIPE_DEPTH) and selector placement against the driver's 20-step line and the 300-step steady state.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/r04_latency_ab.log; : > $L
Q="--no-cpu-baseline --exact-steps 0 --config4-steps 0 --no-decoder-leg"
for rep in 1 2; do
for D in 3 2 1; do
for S in back late; do
  for K in "20 5" "300 20"; do
    set -- $K
    MV_PIPE_DEPTH=$D MV_PIPE_SELECTOR_ON=$S timeout 200 python bench.py --steps $1 --warmup $2 $Q 2>/dev/null | tail -1 > /tmp/line.json
    python - "$D" "$S" "$1" >> $L <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/line.json")); t = d.get("timeline") or {}
    print(f"depth {sys.argv[1]} selector {sys.argv[2]:5s} steps {sys.argv[3]:>3s}: {d['value']:8.1f} fps  {d['ms_per_step']:.4f} ms/step  period {t.get('period_us')} gemm {t.get('gemm_us')} start->pose {t.get('gemm_start_to_pose_us')}")
except Exception as e:
    print("failed", sys.argv[1:], e)
PY
  done
done
done
done
cat $L
