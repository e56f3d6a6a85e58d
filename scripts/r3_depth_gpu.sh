#!/bin/bash
# tracked frames in flight: 3 (default) vs 4 / 5 (libraries built with -DMV_MAX_PENDING=4 / 5)
cd $GRAFT_REPO_ROOT
L=gpurun_out/r3_depth.log; : > $L
run() {
  echo "== $*" >> $L
  env "$@" timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-decoder-leg --exact-steps 0 --config4-steps ${C4:-0} 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'],'fps',d['ms_per_step'],'ms | GEMM',r['avg_launch_us'],'alone',r.get('isolated_avg_launch_us'),'| timeline',d.get('timeline'))
c=d.get('config4')
if c: print('   config4',c['value'],c['ms_per_step'],c['roofline']['avg_launch_us'],c.get('timeline'))
" >> $L 2>&1
}
S=$PWD/profiles/probes
C4=60 run MV_X=0
C4=60 run MACVO_HIP_LIB=$S/libmacvo_hip_depth4.so MV_PIPE_MAX_DEPTH=4 MV_PIPE_DEPTH=4
run MACVO_HIP_LIB=$S/libmacvo_hip_depth5.so MV_PIPE_MAX_DEPTH=5 MV_PIPE_DEPTH=5
run MACVO_HIP_LIB=$S/libmacvo_hip_depth4.so MV_PIPE_MAX_DEPTH=4 MV_PIPE_DEPTH=4 MV_PIPE_VOL_BUFS=2
C4=60 run MV_PIPE_ASYNC_BACKEND=0
for R in 1 2 3; do echo "== f16x2 regions $R" >> $L; MV_SPLIT_REGIONS=$R MV_SPLIT_MODE=f16x2 python tools/kernel_bench.py volume_split --iters 100 2>&1 | grep volume_split >> $L; done
cat $L
