#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline --no-kernel-events"
run() { # label, steps, env...
  lab=$1; st=$2; shift; shift
  env "$@" timeout 200 python bench.py --steps $st --warmup 5 $Q > gpurun_out/r06r_$lab.json 2> gpurun_out/r06r_$lab.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06r_$lab.json") if l.startswith("{")][-1])
print("$lab", "steps", $st, "value", d["value"])
PY
}
for rep in 1 2; do
run base_$rep 300 X=1
run qpb8_$rep 300 MV_LOOKUP_QPB=8
run fin512_$rep 300 MV_KP_FINISH_SMALL_NT=512
run both_$rep 300 MV_LOOKUP_QPB=8 MV_KP_FINISH_SMALL_NT=512
done
run base_20 20 X=1
run qpb8_20 20 MV_LOOKUP_QPB=8
run fin512_20 20 MV_KP_FINISH_SMALL_NT=512
