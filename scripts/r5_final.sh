#!/bin/bash
# round 5, closing evidence (after the pipelined patch embedding became the default): suite, the two driver-style lines, patch-embedding traces + PMC
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r05b_suite.log; tail -3 gpurun_out/r05b_suite.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee gpurun_out/r05b_smoke.log
timeout 600 python bench.py --steps 20 > gpurun_out/r05b_bench_steps20_line.json 2> gpurun_out/r05b_bench_steps20.err; echo "steps20 rc=$?"
timeout 600 python bench.py > gpurun_out/r05b_bench_default_line.json 2> gpurun_out/r05b_bench_default.err; echo "default rc=$?"
python - <<'PY'
import json
for f in ("steps20", "default"):
    d = json.loads([l for l in open(f"gpurun_out/r05b_bench_{f}_line.json") if l.startswith("{")][-1])
    print(f, "value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "pe", d["patch_embed"]["us_per_frame"], d["patch_embed"]["fast_mode"]["us_per_frame"],
          "north_star", d["parity"]["within_north_star"])
PY
OUT=gpurun_out/kprof_r05b; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT" -o trace -- python $R/tools/kernel_bench.py patch_embed --iters 30 ) > $OUT/run.log 2>&1
cp $(find $OUT -name "*kernel_stats.csv" | head -1) gpurun_out/r05b_patch_embed_kernel_stats.csv; grep "cost_patch" $OUT/run.log > gpurun_out/r05b_patch_embed_run.log
( cd /tmp && timeout 300 python $R/tools/kernel_bench.py patch_embed --iters 20 --H 640 --W 640 ) 2>&1 | grep "patch_embed\|cost_patch" >> gpurun_out/r05b_patch_embed_run.log
rm -rf $OUT
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/pmc_r05b_patch_embed_$i -o pmc -- python $R/tools/kernel_bench.py patch_embed --iters 5 ) > gpurun_out/pmc_r05b_pe_$i.log 2>&1
done
python tools/pmc_summary.py r05b_patch_embed "patch_embed" > gpurun_out/pmc_r05b_patch_embed.txt 2>&1
rm -rf gpurun_out/pmc_r05b_patch_embed_[0-9] gpurun_out/pmc_r05b_pe_[0-9].log
cat gpurun_out/r05b_patch_embed_run.log
