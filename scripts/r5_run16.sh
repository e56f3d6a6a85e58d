#!/bin/bash
# round 5: MV_PIPE_LAYOUT=alt — even / odd frames' decoder side (lookups + selector segment) on two streams, backend + solve in order on one: still four queues
cd "${GRAFT_REPO_ROOT:-/root/repo}"
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline"
run() { name=$1; shift
  env "$@" timeout 300 python bench.py $ARGS $Q 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$name', '$ARGS', 'value',d['value'],'ms',d['ms_per_step'],'gemm',d['roofline']['avg_launch_us'],'tl',d['timeline'])"
}
ARGS="--steps 300"
run base_d2
run alt_d2 MV_PIPE_LAYOUT=alt
run alt_d3_v3 MV_PIPE_LAYOUT=alt MV_PIPE_DEPTH=3
run alt_d3_v4 MV_PIPE_LAYOUT=alt MV_PIPE_DEPTH=3 MV_PIPE_VOL_BUFS=4
ARGS="--steps 20"
run base_d2
run alt_d2 MV_PIPE_LAYOUT=alt
run alt_d3_v4 MV_PIPE_LAYOUT=alt MV_PIPE_DEPTH=3 MV_PIPE_VOL_BUFS=4
ARGS="--steps 100 --lanes 2"
run base_l2
run alt_l2 MV_PIPE_LAYOUT=alt MV_PIPE_DEPTH=3 MV_PIPE_VOL_BUFS=4
MV_PIPE_LAYOUT=alt MV_PIPE_DEPTH=3 MV_PIPE_VOL_BUFS=4 timeout 500 python -m pytest tests/test_gpu_native.py tests/test_gpu_lanes.py -q -x -m gpu 2>&1 | tail -4
