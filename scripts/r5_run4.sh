#!/bin/bash
# round 5, call 4: fused backend (2 launches): bitwise test + pipe suites, then A/B of the 20-step line
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_native.py tests/test_gpu_lanes.py tests/test_gpu_backend.py tests/test_gpu_pipeline.py -q -m gpu -x 2>&1 | tail -8
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline"
for f in 0 1 0 1; do
  MV_PIPE_FUSE_BACKEND=$f timeout 300 python bench.py --steps 20 $Q 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('FUSE=$f steps20 value',d['value'],'ms',d['ms_per_step'],'tl',d['timeline'])"
done
for f in 0 1; do
  MV_PIPE_FUSE_BACKEND=$f timeout 300 python bench.py --steps 300 $Q 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('FUSE=$f steps300 value',d['value'],'ms',d['ms_per_step'],'tl',d['timeline'])"
done
