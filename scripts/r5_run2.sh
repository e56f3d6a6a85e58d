#!/bin/bash
# round 5, call 2: upsample (cheaper exp) + 16-bit patch embedding: tests, then kernel traces
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_corr.py tests/test_gpu_patch_embed.py tests/test_gpu_decoder_harness.py tests/test_gpu_flowformer_host.py -q -m gpu -x -k "not end_to_end" 2>&1 | tail -8
for n in 4 8; do MV_UPS_NSX=$n timeout 120 python profiles/probes/r5_upsample_ab.py 2>&1 | grep NSX; done
bash scripts/profile_kernels_gpu.sh r05a_ups_pe upsample patch_embed --iters 30
cat gpurun_out/r05a_ups_pe_kernel_stats.csv | head -12
