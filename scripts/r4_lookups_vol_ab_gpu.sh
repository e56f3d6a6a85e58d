#!/bin/bash
# Round-4 A/B: the 12 lookups behind the GEMM on the GEMM's own stream (serialised, GEMM undisturbed) vs on the decoder-side stream (default).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/r04_lookups_vol_ab.log; : > $L
Q="--no-cpu-baseline --exact-steps 0 --config4-steps 0 --no-decoder-leg"
for rep in 1 2; do
for LK in main serial; do
  for K in "20 5" "300 20"; do
    set -- $K
    MV_PIPE_SERIAL_LOOKUPS=$([ $LK = serial ] && echo 1 || echo 0) timeout 200 python bench.py --steps $1 --warmup $2 $Q 2>/tmp/err.txt | tail -1 > /tmp/line.json; tail -2 /tmp/err.txt | cut -c1-200 >> $L
    python - "$LK" "$1" >> $L <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/line.json")); t = d.get("timeline") or {}; r = d["roofline"]
    print(f"lookups on {sys.argv[1]:4s} steps {sys.argv[2]:>3s}: {d['value']:8.1f} fps  {d['ms_per_step']:.4f} ms/step  gemm events {r['avg_launch_us']} us frac {r['frac']} alone {r.get('isolated_avg_launch_us')} | period {t.get('period_us')} idle {t.get('gemm_stream_idle_us')} start->pose {t.get('gemm_start_to_pose_us')}")
except Exception as e:
    print("failed", sys.argv[1:], e)
PY
  done
done
done
cat $L
