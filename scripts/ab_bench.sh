#!/bin/bash
# A/B driver of bench.py's hot-path line on the GPU box: one run per "label:steps:ENV=VAL,ENV=VAL,..." argument, value + host figures printed per run.
#   gpurun -- 'bash scripts/ab_bench.sh base:300: nolean:300:MV_PIPE_LEAN=0 dd0:20:MV_PIPE_DEVICE_DRAW=0'
# (the round-6 A/B logs under profiles/ — r06_device_draw_ab.log, r06_chain_gaps.log — were produced by one-off copies of this loop; the environment of every line is in the log)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
Q="--exact-steps 0 --config4-steps 0 --fast-mode-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline --no-kernel-events ${AB_EXTRA:-}"
for spec in "$@"; do
  lab=${spec%%:*}; rest=${spec#*:}; st=${rest%%:*}; envs=${rest#*:}
  IFS=',' read -r -a kv <<< "$envs"
  env "${kv[@]:-X=1}" timeout 300 python bench.py --steps "$st" --warmup 5 $Q > "gpurun_out/ab_$lab.json" 2> "gpurun_out/ab_$lab.err"
  python - "$lab" "$st" <<'PY'
import json, sys
lab, st = sys.argv[1:3]
d = json.loads([l for l in open(f"gpurun_out/ab_{lab}.json") if l.startswith("{")][-1])
h = d.get("host") or {}
print(lab, "steps", st, "value", d["value"], "threads", h.get("host_threads"), "issue", h.get("host_issue_us_per_frame"), "wait", h.get("host_flow_control_wait_us_per_frame"))
PY
done
