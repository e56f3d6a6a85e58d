#!/bin/bash
# configs[4] (32 lanes) placement A/B
cd $GRAFT_REPO_ROOT
L=gpurun_out/r3_c4.log; : > $L
run() {
  echo "== $*" >> $L
  env "$@" timeout 300 python bench.py --lanes 32 --steps 100 --warmup 10 --no-cpu-baseline --no-decoder-leg --exact-steps 0 --config4-steps 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'],'fps',d['ms_per_step'],'ms | GEMM',r['avg_launch_us'],'alone',r.get('isolated_avg_launch_us'),'| timeline',d.get('timeline'))
" >> $L 2>&1
}
run MV_X=0
run MV_PIPE_SELECTOR_ON=back
run MV_PIPE_SELECTOR_ON=back MV_PIPE_ASYNC_BACKEND=0
run MV_PIPE_ASYNC_BACKEND=0
run MV_PIPE_VOL_BUFS=2
run MV_PIPE_DEPTH=2
cat $L
