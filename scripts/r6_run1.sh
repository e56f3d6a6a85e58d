#!/bin/bash
# round 6, first device-driven-frame run: the new tests, the driver / lanes suites, then the 20-step line device-driven vs host-drawn
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_device_draw.py tests/test_gpu_native.py tests/test_gpu_lanes.py -q -x 2>&1 | tail -25 > gpurun_out/r06a_suite.log; tail -12 gpurun_out/r06a_suite.log
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline"
for dd in 1 0 1 0; do
  MV_PIPE_DEVICE_DRAW=$dd timeout 200 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/r06a_s20_dd$dd.json 2> gpurun_out/r06a_s20_dd$dd.err; echo "dd=$dd rc=$?"
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06a_s20_dd$dd.json") if l.startswith("{")][-1])
print("dd=$dd steps20 value", d["value"], "ms/step", d["ms_per_step"], "gemm us", d["roofline"]["avg_launch_us"], d.get("timeline"))
PY
done
for dd in 1 0; do
  MV_PIPE_DEVICE_DRAW=$dd timeout 200 python bench.py --steps 300 --warmup 5 $Q > gpurun_out/r06a_s300_dd$dd.json 2> gpurun_out/r06a_s300_dd$dd.err; echo "dd=$dd rc=$?"
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06a_s300_dd$dd.json") if l.startswith("{")][-1])
print("dd=$dd steps300 value", d["value"], "ms/step", d["ms_per_step"], "gemm us", d["roofline"]["avg_launch_us"])
PY
done
