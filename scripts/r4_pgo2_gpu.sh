#!/bin/bash
# PGO kernel iteration: parity (oracle, golden, kernel == twin), alone-times, stamps.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/r04_pgo2.log; : > $L
timeout 300 python -m pytest tests/test_gpu_backend.py tests/test_gpu_golden.py -k "pgo" -x -q 2>&1 | tail -5 >> $L
timeout 120 python tools/kernel_bench.py pgo --iters 100 2>&1 | grep "^pgo" >> $L
echo "== MV_PGO_SPEC=0" >> $L
MV_PGO_SPEC=0 timeout 120 python tools/kernel_bench.py pgo --iters 100 2>&1 | grep "^pgo" | head -2 >> $L
echo "== stamps" >> $L
timeout 120 python profiles/probes/pgo_stamps.py 2>&1 | grep -v amdgpu.ids >> $L
cat $L
