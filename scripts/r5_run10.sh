#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for d in 0 1 2 4 8 3 6 7 15; do
  echo -n "MV_PE_DBG=$d: "; MV_PE_DBG=$d MV_PE_PIPELINED=1 timeout 200 python tools/kernel_bench.py patch_embed --iters 30 2>&1 | grep "cost_patch_embed<bf16" | sed 's/TFLOP.*//' | tr '\n' ' '; echo
done
