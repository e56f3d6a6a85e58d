#!/bin/bash
# round 5, call 5: A/B fused backend x native permutations x timeline events out of the timed region; host breakdown
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline"
run() { # name, env..., args
  name=$1; shift
  env "$@" timeout 300 python bench.py $ARGS $Q 2>gpurun_out/r5_run5_trace_$name.err | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$name', '$ARGS', 'value',d['value'],'ms',d['ms_per_step'],'gemm',d['roofline']['avg_launch_us'],'tl',d['timeline'])"
}
for rep in 1 2; do
ARGS="--steps 20"
run fuse1_native MV_PIPE_FUSE_BACKEND=1
run fuse0_native MV_PIPE_FUSE_BACKEND=0
ARGS="--steps 20 --host-randperm"
run fuse1_hostperm MV_PIPE_FUSE_BACKEND=1
run fuse0_hostperm MV_PIPE_FUSE_BACKEND=0
ARGS="--steps 20"
run fuse1_native_tlin MV_PIPE_FUSE_BACKEND=1 MV_BENCH_TIMELINE_IN_REGION=1
done
ARGS="--steps 300"
run fuse1_native MV_PIPE_FUSE_BACKEND=1 MV_BENCH_TRACE=1
run fuse0_native MV_PIPE_FUSE_BACKEND=0
ARGS="--steps 300 --host-randperm"
run fuse0_hostperm MV_PIPE_FUSE_BACKEND=0
python - <<'PY'
import re
t=open('gpurun_out/r5_run5_trace_fuse1_native.err').read()
m=re.search(r"step-finished times \(us\): ([0-9 ]+)\|", t)
if m:
    xs=[int(x) for x in m.group(1).split()]
    print("first 40 step-finish deltas (us):", [b-a for a,b in zip(xs,xs[1:])])
PY
timeout 120 python tools/host_breakdown.py 400 2>&1 | tail -12
