#!/bin/bash
# round 5: alt layout x CUs left without a GEMM workgroup (MV_SPLIT_FREE_CUS)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline"
run() { name=$1; shift
  env "$@" timeout 300 python bench.py $ARGS $Q 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$name', '$ARGS', 'value',d['value'],'ms',d['ms_per_step'],'gemm',d['roofline']['avg_launch_us'],'tl',d['timeline'])"
}
A="MV_PIPE_LAYOUT=alt MV_PIPE_DEPTH=3"
for ARGS in "--steps 20" "--steps 300"; do
run base_d2
run alt_free0 $A
run alt_free8 $A MV_SPLIT_FREE_CUS=8
run alt_free16 $A MV_SPLIT_FREE_CUS=16
run alt_free32 $A MV_SPLIT_FREE_CUS=32
run base_free16 MV_SPLIT_FREE_CUS=16
done
