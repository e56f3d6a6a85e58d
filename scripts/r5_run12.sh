#!/bin/bash
# round 5: the 512-thread finishing workgroup of the selector (fits beside a GEMM wave: 2 x 106 registers per SIMD) against the 1024-thread one, with the round-5 pipe
cd "${GRAFT_REPO_ROOT:-/root/repo}"
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline"
run() { name=$1; shift
  env "$@" timeout 300 python bench.py $ARGS $Q 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$name', '$ARGS', 'value',d['value'],'ms',d['ms_per_step'],'gemm',d['roofline']['avg_launch_us'],'tl',d['timeline'])"
}
for rep in 1 2 3; do
ARGS="--steps 20"
run nt1024 MV_KP_FINISH_SMALL_NT=1024
run nt512 MV_KP_FINISH_SMALL_NT=512
done
ARGS="--steps 300"
run nt1024 MV_KP_FINISH_SMALL_NT=1024
run nt512 MV_KP_FINISH_SMALL_NT=512
run nt1024 MV_KP_FINISH_SMALL_NT=1024
run nt512 MV_KP_FINISH_SMALL_NT=512
