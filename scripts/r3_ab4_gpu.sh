#!/bin/bash
cd $GRAFT_REPO_ROOT
L=gpurun_out/r3_ab4.log; : > $L
run() {
  echo "== $*" >> $L
  env "$@" timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-decoder-leg --exact-steps 0 --config4-steps 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'],'fps',d['ms_per_step'],'ms | GEMM',r['avg_launch_us'],'alone',r.get('isolated_avg_launch_us'),'| timeline',d.get('timeline'))
" >> $L 2>&1
}
run MV_X=0
run MV_SPLIT_WGS=248
run MV_SPLIT_WGS=240
run MV_SPLIT_WGS=232
run MV_KP_FINISH_SMALL_NT=512
cat $L
