#!/bin/bash
cd $GRAFT_REPO_ROOT
L=gpurun_out/r3_lookup.log; : > $L
timeout 600 python -m pytest tests/test_gpu_corr.py tests/test_gpu_fullsize.py -x -q -k "lookup" 2>&1 | tail -5 >> $L
python tools/kernel_bench.py lookup --iters 100 >> $L 2>&1
python tools/kernel_bench.py lookup --iters 50 --B 64 >> $L 2>&1
python tools/kernel_bench.py lookup --iters 50 --H 720 --W 1280 >> $L 2>&1
timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-decoder-leg --exact-steps 0 --config4-steps 100 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'],'fps',d['ms_per_step'],'ms | GEMM',r['avg_launch_us'],'alone',r.get('isolated_avg_launch_us'),'| timeline',d.get('timeline'))
c=d.get('config4')
if c: print('   config4',c['value'],c['ms_per_step'],c['roofline']['avg_launch_us'],c.get('timeline'))
" >> $L 2>&1
cat $L
