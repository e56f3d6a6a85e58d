#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline --no-kernel-events"
run() { # label, steps, env...
  lab=$1; st=$2; shift; shift
  env "$@" timeout 200 python bench.py --steps $st --warmup 5 $Q > gpurun_out/r06u_$lab.json 2> gpurun_out/r06u_$lab.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06u_$lab.json") if l.startswith("{")][-1])
print("$lab", "steps", $st, "value", d["value"])
PY
}
P=$PWD/mac-vo_amd/csrc/build_probe/libprobe_epint.so
for rep in 1 2 3; do
run base300_$rep 300 X=1
run nt300_$rep 300 MACVO_HIP_LIB=$P
done
for rep in 1 2 3; do
run base20_$rep 20 X=1
run nt20_$rep 20 MACVO_HIP_LIB=$P
done
