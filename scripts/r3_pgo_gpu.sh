#!/bin/bash
cd $GRAFT_REPO_ROOT
L=gpurun_out/r3_pgo.log; : > $L
timeout 900 python -m pytest tests/test_gpu_backend.py tests/test_gpu_golden.py tests/test_gpu_pipeline.py tests/test_gpu_lanes.py -x -q 2>&1 | tail -6 >> $L
python tools/kernel_bench.py pgo --iters 100 2>&1 | grep pgo >> $L
cat $L
