#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/r04_pe5.log; : > $L
timeout 600 python -m pytest tests/test_gpu_patch_embed.py -x -q 2>&1 | tail -8 >> $L
timeout 300 python tools/kernel_bench.py patch_embed --iters 30 2>&1 | grep -v amdgpu.ids >> $L
bash scripts/pmc_gpu.sh r04_patch_embed5 patch_embed 2>&1 | grep -A22 "cost_patch_embed" >> $L
cat $L
