#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
Q="--exact-steps 0 --config4-steps 0 --fast-mode-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline --no-kernel-events --no-ramp"
OUT=gpurun_out/ktrace_r06; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d "$R/$OUT" -o trace -- python $R/bench.py --steps 120 --warmup 5 $Q ) > $OUT/run.log 2>&1
F=$(find $OUT -name "*kernel_trace.csv" | head -1); echo $F; wc -l $F
python - "$F" <<'PY'
import csv, sys, collections, statistics as S
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
def short(n):
    for k in ("corr_lookup", "corr_volume_split", "volume_pack", "kp_nms", "kp_finish", "backend_front", "pgo_solve"):
        if k in n: return k
    return n[:24]
for q, v in byq.items():
    v.sort()
    v = v[len(v) // 2:]            # the timed half
    names = collections.Counter(short(n) for _, _, n in v)
    gaps = collections.defaultdict(list)
    durs = collections.defaultdict(list)
    for (s0, e0, n0), (s1, e1, n1) in zip(v, v[1:]):
        gaps[(short(n0), short(n1))].append((s1 - e0) / 1e3)
    for s0, e0, n0 in v:
        durs[short(n0)].append((e0 - s0) / 1e3)
    print("queue", q, dict(names.most_common(5)))
    for k, g in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:8]:
        print("    gap %-28s n=%4d median %7.2f us  p10 %7.2f  p90 %7.2f" % (" -> ".join(k), len(g), S.median(g), sorted(g)[len(g) // 10], sorted(g)[len(g) * 9 // 10]))
    for k, d in durs.items():
        print("    dur %-20s median %7.2f us" % (k, S.median(d)))
PY
rm -rf $OUT
