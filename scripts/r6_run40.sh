#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline"
run() { lab=$1; shift
  OUT=gpurun_out/kp_$lab; rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && env "$@" timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT" -o trace -- python $R/bench.py --steps 20 --warmup 5 $Q ) > $OUT/run.log 2>&1
  F=$(find $OUT -name "*kernel_stats.csv" | head -1)
  python - "$F" "$lab" "$OUT/run.log" <<'PY'
import csv, sys, json
rows = list(csv.reader(open(sys.argv[1])))[1:8]
d = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
print(sys.argv[2], "value(traced)", d["value"], " ".join("%s %.1f" % (r[0].split("::")[-1][:14], float(r[3]) / 1e3) for r in rows))
PY
  rm -rf $OUT
}
run lean_async X=1
run lean_inline MV_PIPE_ASYNC_BACKEND=0
run nolean_async MV_PIPE_LEAN=0
run nolean_inline MV_PIPE_LEAN=0 MV_PIPE_ASYNC_BACKEND=0
run lean_async_b X=1
run lean_inline_b MV_PIPE_ASYNC_BACKEND=0
