#!/bin/bash
# Where do the waves of corr_volume_split_stream spend their cycles?  SQ wait / active counters + effective clock (GRBM_GUI_ACTIVE / wall
# time) for the production kernel, knock-out builds (profiles/probes/split_variants.sh) and zero operands.  PMC passes only.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
OUT=gpurun_out/r03_split_waitpmc.log; : > $OUT
( cd /tmp && rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u | tr '\n' ' ' ) > gpurun_out/r03_sq_counters.txt
run() {   # tag, lib, mode, extra args
  local tag=$1 lib=$2 mode=$3; shift 3
  local i=0
  for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_FLAT"; do
    i=$((i+1))
    ( cd /tmp && MACVO_HIP_LIB=$lib MV_SPLIT_MODE=$mode timeout 120 rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/wpmc_${tag}_$i -o pmc -- python $R/tools/kernel_bench.py volume_split --iters 5 "$@" ) > gpurun_out/wpmc_${tag}_$i.log 2>&1
    grep "volume_split" gpurun_out/wpmc_${tag}_$i.log | tail -1 | sed "s/^/[$tag pass $i] /" >> $OUT
  done
}
P=$R/mac-vo_amd/libmacvo_hip.so
S=$R/profiles/probes
run f16x2 $P f16x2
run f16x2_zeros $P f16x2 --zeros
run f16x2_k1 $S/libmacvo_hip_split_k1.so f16x2
run f16x2_k15 $S/libmacvo_hip_split_k15.so f16x2
run bf16x3 $P bf16x3
python - >> $OUT <<'PY'
import collections, csv, glob, re
for tag in ("f16x2", "f16x2_zeros", "f16x2_k1", "f16x2_k15", "bf16x3"):
    agg = collections.defaultdict(list); dur = []
    for f in glob.glob(f"gpurun_out/wpmc_{tag}_[0-9]/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "split_stream" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    print(f"== {tag}")
    for k in sorted(m): print(f"   {k:32s} {m[k]:.4e}")
    if "GRBM_GUI_ACTIVE" in m and "SQ_WAVE_CYCLES" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        wc = m["SQ_WAVE_CYCLES"] * 4 / 1024
        print(f"   kernel cycles {cyc:.0f}; per-wave cycles {wc:.0f}; MFMA busy {m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024/cyc:.3f}; "
              f"wave time: parked {m.get('SQ_WAIT_ANY',0)/m['SQ_WAVE_CYCLES']:.3f} issue-stall {m.get('SQ_WAIT_INST_ANY',0)/m['SQ_WAVE_CYCLES']:.3f} active {m.get('SQ_ACTIVE_INST_ANY',0)/m['SQ_WAVE_CYCLES']:.3f}")
PY
cat $OUT
