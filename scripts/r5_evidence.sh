#!/bin/bash
# round 5 evidence (one gpurun call): full GPU suite, smoke, the driver's lines, rocprofv3 kernel traces, PMC passes (counters only, no trace domains mixed in)
# usage: scripts/r5_evidence.sh [suite|lines|traces|pmc ...]   (default: everything)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
WHAT="${*:-suite lines traces pmc}"
has() { case " $WHAT " in *" $1 "*) return 0;; *) return 1;; esac; }
Q="--exact-steps 0 --config4-steps 0 --no-decoder-leg --end-to-end-frames 0 --plugin-frames 0 --no-cpu-baseline"
if has suite; then
  timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r05_suite.log; tail -4 gpurun_out/r05_suite.log
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/r05_smoke.log
fi
if has lines; then
  timeout 600 python bench.py --steps 20 > gpurun_out/r05_bench_steps20_line.json 2> gpurun_out/r05_bench_steps20.err; echo "steps20 rc=$?"
  timeout 600 python bench.py > gpurun_out/r05_bench_default_line.json 2> gpurun_out/r05_bench_default.err; echo "default rc=$?"
  timeout 400 python bench.py --steps 100 --height 720 --width 1280 --feat-dtype f16 --layout hwc --volume-store encoder --exact-steps 0 --config4-steps 0 --end-to-end-frames 0 \
      --plugin-frames 0 --cpu-frames 2 --parity-frames 2 --reference-frames 0 --pool 6 > gpurun_out/r05_bench_720p_f16hwc_enc16_line.json 2> gpurun_out/r05_bench_720p.err; echo "720p rc=$?"
  timeout 400 python bench.py --steps 100 --feat-dtype f16 --layout hwc --volume-store encoder --exact-steps 0 --config4-steps 0 --end-to-end-frames 0 --plugin-frames 0 \
      --cpu-frames 2 --parity-frames 2 --reference-frames 0 > gpurun_out/r05_bench_480p_f16hwc_enc16_line.json 2> gpurun_out/r05_bench_480p_enc16.err; echo "480p enc16 rc=$?"
  python - <<'PY'
import json
for f in ("steps20", "default", "720p_f16hwc_enc16", "480p_f16hwc_enc16"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r05_bench_{f}_line.json") if l.startswith("{")][-1])
        print(f, "value", d["value"], "ms/step", d["ms_per_step"], "roofline", {k: d["roofline"].get(k) for k in ("kernel", "avg_launch_us", "frac", "isolated_avg_launch_us")},
              "within_north_star", (d.get("parity") or {}).get("within_north_star"))
    except Exception as e:
        print(f, "FAILED", repr(e)[:200])
PY
fi
trace() { # tag, command...
  TAG=$1; shift
  OUT=gpurun_out/kprof_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT" -o trace -- "$@" ) > "$OUT/run.log" 2>&1
  f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "gpurun_out/${TAG}_kernel_stats.csv"
  grep -v "^W2\|rocprofv3\|amdgpu.ids\|^E2\|^I2" "$OUT/run.log" | tail -14 > "gpurun_out/${TAG}_run.log"
  rm -rf "$OUT"
}
if has traces; then
  trace r05_bench python $R/bench.py --steps 100 --warmup 10 $Q
  trace r05_kernels python $R/tools/kernel_bench.py upsample patch_embed select cov pgo lookup volume_split volume_f16 --iters 30
  trace r05_cfg2_720p_kernels python $R/tools/kernel_bench.py volume_f16 lookup upsample patch_embed --H 720 --W 1280 --iters 10
  MV_PE_STRIP=1 trace r05_pe_strip_480p python $R/tools/kernel_bench.py patch_embed --iters 20
  head -14 gpurun_out/r05_bench_kernel_stats.csv | cut -c1-150
fi
pmc() { # tag, kernel_bench args
  TAG=$1; shift
  i=0
  for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" \
             "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    ( cd /tmp && timeout 300 rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -o pmc -- python $R/tools/kernel_bench.py "$@" --iters 5 ) > gpurun_out/pmc_${TAG}_$i.log 2>&1
  done
  python tools/pmc_summary.py "$TAG" "$*" > gpurun_out/pmc_${TAG}.txt 2>&1
  rm -rf gpurun_out/pmc_${TAG}_[0-9] gpurun_out/pmc_${TAG}_[0-9].log
  grep -c derived gpurun_out/pmc_${TAG}.txt
}
if has pmc; then
  pmc r05_upsample upsample
  pmc r05_patch_embed patch_embed
  pmc r05_corr_volume_split_f16x2 volume_split
  pmc r05_cfg2_720p volume_f16 lookup upsample --H 720 --W 1280
  pmc r05_patch_embed_720p patch_embed --H 720 --W 1280
  grep -A3 "upsample\|patch_embed" gpurun_out/pmc_r05_upsample.txt gpurun_out/pmc_r05_patch_embed.txt | grep "derived" | cut -c1-400
fi
