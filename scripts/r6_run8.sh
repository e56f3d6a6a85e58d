#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
run() { # label, env..., -- extra
  lab=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 400 python bench.py --steps 20 --warmup 5 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --exact-steps 0 "$@" > gpurun_out/r06h_$lab.json 2> gpurun_out/r06h_$lab.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06h_$lab.json") if l.startswith("{")][-1])
c = d.get("config4") or {}
print("$lab", "value", d["value"], "config4", c.get("value"), (c.get("timeline") or {}).get("period_us"))
PY
}
run dd0_full MV_PIPE_DEVICE_DRAW=0 --
run dd1_cpu1_ref0 X=1 -- --cpu-frames 1 --parity-frames 2 --reference-frames 0
run dd1_omp1 OMP_NUM_THREADS=1 --
run dd1_async MV_PIPE_ASYNC_BACKEND=1 --
run dd1_a6 MV_PIPE_DD_AHEAD=6 --
