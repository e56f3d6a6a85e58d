#!/bin/bash
cd $GRAFT_REPO_ROOT
L=gpurun_out/r3_late2.log; : > $L
run() {
  env "$@" timeout 120 python bench.py --steps ${K:-300} --warmup ${W:-20} --no-cpu-baseline --no-decoder-leg --exact-steps 0 --config4-steps 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d.get('timeline') or {}
print('K=${K:-300}', '$*', d['value'],'fps | period',t.get('period_us'),'latency',t.get('gemm_start_to_pose_us'))
" >> $L 2>&1
}
for i in 1 2 3; do run MV_PIPE_SELECTOR_ON=back; run MV_PIPE_SELECTOR_ON=late; done
for i in 1 2; do K=20 W=5 run MV_PIPE_SELECTOR_ON=back; K=20 W=5 run MV_PIPE_SELECTOR_ON=late; done
cat $L
MV_PIPE_SELECTOR_ON=late timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 >> $L
tail -3 $L
