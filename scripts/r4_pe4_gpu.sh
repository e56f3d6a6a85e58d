#!/bin/bash
# Round-4: patch embed with conv1 on 16x16x32 tiles + symmetric K-half hand-over; then the whole GPU suite.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/r04_pe4.log; : > $L
timeout 600 python -m pytest tests/test_gpu_patch_embed.py -x -q 2>&1 | tail -15 >> $L
timeout 300 python tools/kernel_bench.py patch_embed --iters 30 2>&1 | grep -v amdgpu.ids >> $L
bash scripts/pmc_gpu.sh r04_patch_embed3 patch_embed 2>&1 | grep -A22 "cost_patch_embed" >> $L
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 >> $L
cat $L
