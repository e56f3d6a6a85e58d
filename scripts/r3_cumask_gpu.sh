#!/bin/bash
# spatial partitioning A/B with the persistent f16x2 GEMM: fewer persistent workgroups, CU masks for the streams
cd $GRAFT_REPO_ROOT
L=gpurun_out/r3_cumask.log; : > $L
run() {
  echo "== $*" >> $L
  env "$@" timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-decoder-leg --exact-steps 0 --config4-steps 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'],'fps',d['ms_per_step'],'ms | GEMM',r['avg_launch_us'],'alone',r.get('isolated_avg_launch_us'),'| timeline',d.get('timeline'))
" >> $L 2>&1
}
run MV_X=0
run MV_SPLIT_WGS=224
run MV_SPLIT_WGS=192
run MV_PIPE_SMALL_CUS=32 MV_SPLIT_WGS=224
run MV_PIPE_SMALL_CUS=64 MV_SPLIT_WGS=192
run MV_PIPE_SMALL_CUS=64 MV_SPLIT_WGS=192 MV_PIPE_CU_INTERLEAVE=1
run MV_PIPE_SMALL_CUS=32 MV_SPLIT_WGS=224 MV_PIPE_CU_INTERLEAVE=1
cat $L
