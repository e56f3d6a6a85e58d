#!/bin/bash
# rocprofv3 kernel stats of the 32-lane pipe under environment variants (probe, round 6):  scripts/c4_trace.sh "label:ENV=V,ENV=V" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
Q="--exact-steps 0 --config4-steps 0 --fast-mode-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline --parity-frames 0 --no-kernel-events"
for spec in "$@"; do
  lab=${spec%%:*}; envs=${spec#*:}
  IFS=',' read -r -a kv <<< "$envs"
  OUT=gpurun_out/c4prof_$lab; rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && env "${kv[@]:-X=1}" timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT" -o trace -- python $R/bench.py --lanes ${C4_LANES:-32} --steps 40 --warmup 5 $Q ) > $OUT/run.log 2>&1
  cp $(find $OUT -name "*kernel_stats.csv" | head -1) gpurun_out/c4_kernel_stats_$lab.csv
  grep '^{' $OUT/run.log | tail -1 > gpurun_out/c4_line_$lab.json
  python - "$lab" <<'PY'
import csv, json, sys
lab = sys.argv[1]
d = json.loads(open(f"gpurun_out/c4_line_{lab}.json").read())
print("==", lab, "value", d["value"])
for r in csv.DictReader(open(f"gpurun_out/c4_kernel_stats_{lab}.csv")):
    n = r["Name"].split("(anonymous namespace)::")[1][:34] if "(anonymous namespace)::" in r["Name"] else r["Name"][:34]
    if float(r["Percentage"]) > 0.5:
        print(f"   {n:36s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']:>6s}%  min {float(r['MinNs'])/1e3:.1f} max {float(r['MaxNs'])/1e3:.1f}")
PY
  rm -rf $OUT
done
