#!/bin/bash
# PGO: the fused first trial (build at the trial pose) against the unfused build of the same source (-DMV_PGO_FUSED_BUILD=0), parity, stamps.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/r04_pgo3.log; : > $L
timeout 300 python -m pytest tests/test_gpu_backend.py tests/test_gpu_golden.py -k "pgo" -x -q 2>&1 | tail -3 >> $L
for rep in 1 2; do
  echo "== fused (default)" >> $L; timeout 120 python tools/kernel_bench.py pgo --iters 100 2>&1 | grep "^pgo" | head -6 >> $L
  echo "== -DMV_PGO_FUSED_BUILD=0" >> $L; MACVO_HIP_LIB=$PWD/profiles/probes/libmacvo_hip_pgo_nofuse.so timeout 120 python tools/kernel_bench.py pgo --iters 100 2>&1 | grep "^pgo" | head -6 >> $L
done
echo "== stamps" >> $L
timeout 120 python profiles/probes/pgo_stamps.py 2>&1 | grep -v amdgpu.ids >> $L
cat $L
