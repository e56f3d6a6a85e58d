#!/bin/bash
# round 5: frame f's lookups on decoder-side stream f % n (MV_PIPE_LOOKUP_STREAMS) x pipeline depth x volume buffers, one lane
cd "${GRAFT_REPO_ROOT:-/root/repo}"
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline"
run() { name=$1; shift
  env "$@" timeout 300 python bench.py $ARGS $Q 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$name', '$ARGS', 'value',d['value'],'ms',d['ms_per_step'],'gemm',d['roofline']['avg_launch_us'],'tl',d['timeline'])"
}
MP4=MACVO_HIP_LIB=$PWD/profiles/probes/libmacvo_hip_mp4.so
ARGS="--steps 300"
run lk1_d2 MV_PIPE_LOOKUP_STREAMS=1
run lk2_d2 MV_PIPE_LOOKUP_STREAMS=2
run lk2_d3_v3 MV_PIPE_LOOKUP_STREAMS=2 MV_PIPE_DEPTH=3
run lk2_d3_v4 MV_PIPE_LOOKUP_STREAMS=2 MV_PIPE_DEPTH=3 MV_PIPE_VOL_BUFS=4
run lk3_d3_v4_q8 MV_PIPE_LOOKUP_STREAMS=3 MV_PIPE_DEPTH=3 MV_PIPE_VOL_BUFS=4 GPU_MAX_HW_QUEUES=8
run lk3_d3_v4 MV_PIPE_LOOKUP_STREAMS=3 MV_PIPE_DEPTH=3 MV_PIPE_VOL_BUFS=4
run lk2_d4_v5 $MP4 MV_PIPE_LOOKUP_STREAMS=2 MV_PIPE_DEPTH=4 MV_PIPE_MAX_DEPTH=4 MV_PIPE_VOL_BUFS=5
run lk3_d4_v5_q8 $MP4 MV_PIPE_LOOKUP_STREAMS=3 MV_PIPE_DEPTH=4 MV_PIPE_MAX_DEPTH=4 MV_PIPE_VOL_BUFS=5 GPU_MAX_HW_QUEUES=8
ARGS="--steps 20"
run lk1_d2 MV_PIPE_LOOKUP_STREAMS=1
run lk2_d3_v4 MV_PIPE_LOOKUP_STREAMS=2 MV_PIPE_DEPTH=3 MV_PIPE_VOL_BUFS=4
run lk3_d3_v4_q8 MV_PIPE_LOOKUP_STREAMS=3 MV_PIPE_DEPTH=3 MV_PIPE_VOL_BUFS=4 GPU_MAX_HW_QUEUES=8
MV_PIPE_LOOKUP_STREAMS=2 MV_PIPE_DEPTH=3 MV_PIPE_VOL_BUFS=4 timeout 400 python -m pytest tests/test_gpu_native.py -q -x -m gpu 2>&1 | tail -4
