#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
lscpu | grep -E "Model name|Socket|NUMA|Core|Thread|^CPU\(s\)" | head -12
for d in /sys/class/drm/card*/device; do echo "$d numa=$(cat $d/numa_node 2>/dev/null) local_cpulist=$(cat $d/local_cpulist 2>/dev/null)"; done 2>/dev/null | head -4
python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print({k: getattr(p, k) for k in dir(p) if k.startswith("pci")})
PY
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline --no-kernel-events"
run() { # label, steps, env...
  lab=$1; st=$2; shift; shift
  env "$@" timeout 200 python bench.py --steps $st --warmup 5 $Q > gpurun_out/r06o_$lab.json 2> gpurun_out/r06o_$lab.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06o_$lab.json") if l.startswith("{")][-1])
h = d.get("host") or {}
print("$lab", "steps", $st, "value", d["value"], "issue", h.get("host_issue_us_per_frame"), "wait", h.get("host_flow_control_wait_us_per_frame"))
PY
}
export MV_PIPE_FRONT_ON=decoder
for c in 0 3 32 64 96 128 160 200 255; do run core$c 20 taskset -c $c; done
run cores0_7 20 taskset -c 0-7
run cores0_63 20 taskset -c 0-63
run cores64_127 20 taskset -c 64-127
run unpinned 20 X=1
run core0_300 300 taskset -c 0
run unpinned_300 300 X=1
run dd0_core01 20 MV_PIPE_DEVICE_DRAW=0 taskset -c 0,1
run dd0_cores0_7 20 MV_PIPE_DEVICE_DRAW=0 taskset -c 0-7
