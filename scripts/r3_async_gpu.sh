#!/bin/bash
# async backend: parity tests + A/B of the one-lane stream and config4
cd $GRAFT_REPO_ROOT
L=gpurun_out/r3_async.log; : > $L
timeout 600 python -m pytest tests/test_gpu_lanes.py tests/test_gpu_native.py tests/test_gpu_visual_map.py -x -q 2>&1 | tail -8 >> $L
for A in 0 1; do
  echo "== MV_PIPE_ASYNC_BACKEND=$A" >> $L
  MV_PIPE_ASYNC_BACKEND=$A timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-decoder-leg --exact-steps 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'],'fps',d['ms_per_step'],'ms | GEMM',r['avg_launch_us'],'alone',r.get('isolated_avg_launch_us'),'| timeline',d.get('timeline'))
c=d.get('config4')
if c: print('   config4',c['value'],c['ms_per_step'],c['roofline']['avg_launch_us'],c.get('timeline'))
" >> $L 2>&1
done
MV_PIPE_ASYNC_BACKEND=1 timeout 120 python tools/host_breakdown.py 2>&1 | grep -v "^Exception\|^Traceback\|^  File\|AttributeError" | tail -12 >> $L
cat $L
