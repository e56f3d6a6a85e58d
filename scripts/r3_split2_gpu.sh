#!/bin/bash
cd $GRAFT_REPO_ROOT
L=gpurun_out/r3_split2.log; : > $L
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_native.py -x -q 2>&1 | tail -6 >> $L
for M in f16x2 bf16x3; do MV_SPLIT_MODE=$M python tools/kernel_bench.py volume_split --iters 100 2>&1 | grep volume_split >> $L; done
MV_SPLIT_MODE=f16x2 python tools/kernel_bench.py volume_split --iters 100 --zeros 2>&1 | grep volume_split >> $L
MV_SPLIT_MODE=f16x2 python tools/kernel_bench.py volume_split --iters 30 --B 64 2>&1 | grep volume_split >> $L
MV_SPLIT_MODE=f16x2 python tools/kernel_bench.py volume_split --iters 30 --H 720 --W 1280 2>&1 | grep volume_split >> $L
cat $L
