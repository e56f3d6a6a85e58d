#!/bin/bash
# SQ wait counters of the production f16x2 split kernel only (one pass)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
( cd /tmp && MV_SPLIT_MODE=f16x2 timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/wq_f16x2 -o pmc -- python $R/tools/kernel_bench.py volume_split --iters 5 ) > gpurun_out/wq_f16x2.log 2>&1
python - <<'PY'
import collections, csv, glob
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/wq_f16x2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "split_stream" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
cyc = m["GRBM_GUI_ACTIVE"] / 8; wc = m["SQ_WAVE_CYCLES"] * 4 / 1024
print(f"kernel cycles {cyc:.0f}; per-wave cycles {wc:.0f}; MFMA busy {m['SQ_VALU_MFMA_BUSY_CYCLES']/1024/cyc:.3f}; parked {m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:.3f} = {m['SQ_WAIT_ANY']*4/1024:.0f} cyc; issue-stall {m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']:.3f}; active {m['SQ_ACTIVE_INST_ANY']/m['SQ_WAVE_CYCLES']:.3f}")
PY
