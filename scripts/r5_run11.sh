#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_patch_embed.py -q -m gpu 2>&1 | tail -4
timeout 300 python tools/kernel_bench.py patch_embed --iters 15 --H 720 --W 1280 2>&1 | grep "cost_patch"
MV_PE_STRIP=1 timeout 300 python tools/kernel_bench.py patch_embed --iters 30 2>&1 | grep "cost_patch"
