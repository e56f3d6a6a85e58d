#!/bin/bash
cd $GRAFT_REPO_ROOT
L=gpurun_out/r3_ab3.log; : > $L
run() {
  echo "== $*" >> $L
  env "$@" timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-decoder-leg --exact-steps 0 --config4-steps 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'],'fps',d['ms_per_step'],'ms | GEMM',r['avg_launch_us'],'alone',r.get('isolated_avg_launch_us'),'| timeline',d.get('timeline'))
" >> $L 2>&1
}
run MV_X=0
run MV_PIPE_PACK_ON=side
run MV_X=1
run MV_PIPE_PACK_ON=side
cat $L
