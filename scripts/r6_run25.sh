#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for B in 2 64; do
echo "== lookup B=$B default"; python tools/kernel_bench.py lookup --iters 30 --B $B 2>&1 | grep -E "^lookup"
echo "== lookup B=$B forced small variant (2 q/wave, 16 q/WG)"; MV_LOOKUP_SMALL=100000000 python tools/kernel_bench.py lookup --iters 30 --B $B 2>&1 | grep -E "^lookup"
echo "== lookup B=$B forced large variant (4 q/wave, 32 q/WG)"; MV_LOOKUP_SMALL=0 python tools/kernel_bench.py lookup --iters 30 --B $B 2>&1 | grep -E "^lookup"
echo "== lookup B=$B QPB=8"; MV_LOOKUP_QPB=8 python tools/kernel_bench.py lookup --iters 30 --B $B 2>&1 | grep -E "^lookup"
done
