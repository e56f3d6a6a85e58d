#!/bin/bash
cd $GRAFT_REPO_ROOT
L=gpurun_out/r3_late.log; : > $L
run() {
  echo "== K=${K:-300} $*" >> $L
  env "$@" timeout 200 python bench.py --steps ${K:-300} --warmup ${W:-20} --no-cpu-baseline --no-decoder-leg --exact-steps 0 --config4-steps 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'],'fps',d['ms_per_step'],'ms | GEMM',r['avg_launch_us'],'| timeline',json.dumps(d.get('timeline')))
" >> $L 2>&1
}
run MV_X=0
run MV_PIPE_SELECTOR_ON=late
K=20 W=5 run MV_X=0
K=20 W=5 run MV_PIPE_SELECTOR_ON=late
K=20 W=5 run MV_X=0
K=20 W=5 run MV_PIPE_SELECTOR_ON=late
MV_PIPE_SELECTOR_ON=late timeout 600 python -m pytest tests/test_gpu_native.py tests/test_gpu_lanes.py tests/test_gpu_visual_map.py -x -q 2>&1 | tail -3 >> $L
cat $L
