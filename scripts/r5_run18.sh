#!/bin/bash
# round 5: alt layout with independent selector segments (MV_PIPE_ALT_INDEP, default on) against ordered segments and the default layout
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
run alt_d3_ord MV_PIPE_LAYOUT=alt MV_PIPE_DEPTH=3 MV_PIPE_ALT_INDEP=0
run alt_d3_ind MV_PIPE_LAYOUT=alt MV_PIPE_DEPTH=3
run alt_d2_ind MV_PIPE_LAYOUT=alt MV_PIPE_DEPTH=2
ARGS="--steps 300"
run alt_d3_ord MV_PIPE_LAYOUT=alt MV_PIPE_DEPTH=3 MV_PIPE_ALT_INDEP=0
run alt_d3_ind MV_PIPE_LAYOUT=alt MV_PIPE_DEPTH=3
done
MV_PIPE_LAYOUT=alt MV_PIPE_DEPTH=3 timeout 500 python -m pytest tests/test_gpu_native.py tests/test_gpu_lanes.py -q -x -m gpu 2>&1 | tail -4
