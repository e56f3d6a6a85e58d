#!/bin/bash
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_patch_embed.py -q -m gpu -k "pipelined" 2>&1 | tail -8
MV_PE_PIPELINED=1 timeout 300 python tools/kernel_bench.py patch_embed --iters 30 2>&1 | grep -v "amdgpu.ids\|unfused"
MV_PE_PIPELINED=0 timeout 300 python tools/kernel_bench.py patch_embed --iters 30 2>&1 | grep -v "amdgpu.ids\|unfused"
