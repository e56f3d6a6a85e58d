#!/bin/bash
# round 5: alt layout, more CUs without a GEMM workgroup; volume buffers; depth; two lanes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline"
run() { name=$1; shift
  env "$@" timeout 300 python bench.py $ARGS $Q 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$name', '$ARGS', 'value',d['value'],'ms',d['ms_per_step'],'gemm',d['roofline']['avg_launch_us'],'tl',d['timeline'])"
}
A="MV_PIPE_LAYOUT=alt MV_PIPE_DEPTH=3"
for ARGS in "--steps 20" "--steps 300"; do
run alt_free32 $A MV_SPLIT_FREE_CUS=32
run alt_free48 $A MV_SPLIT_FREE_CUS=48
run alt_free64 $A MV_SPLIT_FREE_CUS=64
run alt_free32_v4 $A MV_SPLIT_FREE_CUS=32 MV_PIPE_VOL_BUFS=4
run alt_free32_d2 MV_PIPE_LAYOUT=alt MV_PIPE_DEPTH=2 MV_SPLIT_FREE_CUS=32
done
ARGS="--steps 100 --lanes 2"
run base_l2
run alt_l2_free32 $A MV_SPLIT_FREE_CUS=32
run alt_l2_free0 $A
