#!/bin/bash
# Round-4: patch-embed after software pipelining, Fast-mode (out16 / vol16) tests, then the GPU suite.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/r04_pe2.log; : > $L
timeout 600 python -m pytest tests/test_gpu_patch_embed.py tests/test_gpu_fastmode.py tests/test_gpu_split.py -x -q 2>&1 | tail -15 >> $L
timeout 300 python tools/kernel_bench.py patch_embed --iters 30 2>&1 | grep -v amdgpu.ids >> $L
timeout 300 python tools/kernel_bench.py volume_f16 2>&1 | grep -v amdgpu.ids >> $L
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 >> $L
cat $L
