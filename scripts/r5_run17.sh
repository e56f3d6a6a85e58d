#!/bin/bash
# round 5: MV_PIPE_LAYOUT=alt against the default layout on one box (stream creation order restored), two repetitions
cd "${GRAFT_REPO_ROOT:-/root/repo}"
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline"
run() { name=$1; shift
  env "$@" timeout 300 python bench.py $ARGS $Q 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$name', '$ARGS', 'value',d['value'],'ms',d['ms_per_step'],'gemm',d['roofline']['avg_launch_us'],'tl',d['timeline'])"
}
for rep in 1 2; do
ARGS="--steps 20"
run base_d2
run alt_d3_v4 MV_PIPE_LAYOUT=alt MV_PIPE_DEPTH=3 MV_PIPE_VOL_BUFS=4
run alt_d3_v3 MV_PIPE_LAYOUT=alt MV_PIPE_DEPTH=3
ARGS="--steps 300"
run base_d2
run alt_d3_v4 MV_PIPE_LAYOUT=alt MV_PIPE_DEPTH=3 MV_PIPE_VOL_BUFS=4
done
ARGS="--steps 300"
run base_d3 MV_PIPE_DEPTH=3
run alt_d3_v5 MV_PIPE_LAYOUT=alt MV_PIPE_DEPTH=3 MV_PIPE_VOL_BUFS=5
