#!/bin/bash
# round 5, call 7: strip-mined patch embedding (80x80, 90x160, forced 60x80) + conv1 ky remap in the whole-slice kernel: tests, then timings; launch-thread spin A/B
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1200 python -m pytest tests/test_gpu_patch_embed.py -q -m gpu -x 2>&1 | tail -12
timeout 300 python tools/kernel_bench.py patch_embed --iters 30 2>&1 | grep -v amdgpu.ids
MV_PE_STRIP=1 timeout 300 python tools/kernel_bench.py patch_embed --iters 30 2>&1 | grep -v amdgpu.ids | grep -v unfused
timeout 300 python tools/kernel_bench.py patch_embed --iters 15 --H 720 --W 1280 2>&1 | grep -v amdgpu.ids
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline"
run() { name=$1; shift
  env "$@" timeout 300 python bench.py $ARGS $Q 2>gpurun_out/r5_run7_$name.err | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$name', '$ARGS', 'value',d['value'],'ms',d['ms_per_step'],'gemm',d['roofline']['avg_launch_us'],'tl',d['timeline'])"
  grep "host stats" gpurun_out/r5_run7_$name.err | tail -2
}
for rep in 1 2; do
ARGS="--steps 20"
run spin250 MV_PIPE_HOST_STATS=1
run spin0 MV_PIPE_HOST_STATS=1 MV_PIPE_LAUNCH_SPIN_US=0
done
ARGS="--steps 300"
run spin250 MV_PIPE_HOST_STATS=1
run spin0 MV_PIPE_HOST_STATS=1 MV_PIPE_LAUNCH_SPIN_US=0
