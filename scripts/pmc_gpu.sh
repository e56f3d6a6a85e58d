#!/bin/bash
# PMC passes (counters only, no trace domains mixed in) over tools/kernel_bench.py; usage: scripts/pmc_gpu.sh <tag> <kernel_bench args>
set -u
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd /tmp && rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -o pmc -- python $R/tools/kernel_bench.py "$@" --iters 5 ) > gpurun_out/pmc_${TAG}_$i.log 2>&1
done
python tools/pmc_summary.py "$TAG" "$*"
