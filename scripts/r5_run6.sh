#!/bin/bash
# round 5, call 6: fused backend without fences (filters in the solve's prologue): tests + A/B; depth 3
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_native.py tests/test_gpu_lanes.py tests/test_gpu_backend.py -q -m gpu -x 2>&1 | tail -5
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline"
run() { name=$1; shift
  env "$@" timeout 300 python bench.py $ARGS $Q 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$name', '$ARGS', 'value',d['value'],'ms',d['ms_per_step'],'gemm',d['roofline']['avg_launch_us'],'tl',d['timeline'])"
}
for rep in 1 2 3; do
ARGS="--steps 20"
run fuse1 MV_PIPE_FUSE_BACKEND=1
run fuse0 MV_PIPE_FUSE_BACKEND=0
done
ARGS="--steps 20"
run fuse1_depth3 MV_PIPE_FUSE_BACKEND=1 MV_PIPE_DEPTH=3
run fuse0_depth3 MV_PIPE_FUSE_BACKEND=0 MV_PIPE_DEPTH=3
ARGS="--steps 300"
run fuse1 MV_PIPE_FUSE_BACKEND=1
run fuse0 MV_PIPE_FUSE_BACKEND=0
run fuse1_depth3 MV_PIPE_FUSE_BACKEND=1 MV_PIPE_DEPTH=3
run fuse0_depth3 MV_PIPE_FUSE_BACKEND=0 MV_PIPE_DEPTH=3
