#!/bin/bash
# Round-4 A/B of the split-GEMM variants (profiles/probes/r4_split_variants.sh) + the new bench legs + the GPU suite on the default build.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/r04_split_ab.log; : > $L
P=$PWD/profiles/probes
for rep in 1 2; do
  for v in r3 base imm vacc both; do
    MACVO_HIP_LIB=$P/libmacvo_hip_r4_$v.so timeout 120 python tools/split_ab.py >> $L 2>&1
  done
done
for v in r3 both; do MACVO_HIP_LIB=$P/libmacvo_hip_r4_$v.so timeout 120 python tools/split_ab.py --zeros >> $L 2>&1; done
for v in r3 both; do MACVO_HIP_LIB=$P/libmacvo_hip_r4_$v.so timeout 120 python tools/split_ab.py --B 64 --launches 20 >> $L 2>&1; done
for v in r3 both; do MACVO_HIP_LIB=$P/libmacvo_hip_r4_$v.so timeout 120 python tools/split_ab.py --mode bf16x3 >> $L 2>&1; done
# correctness of the candidate default on the split test file, then in-pipe numbers
MACVO_HIP_LIB=$P/libmacvo_hip_r4_both.so timeout 600 python -m pytest tests/test_gpu_split.py -x -q 2>&1 | tail -4 >> $L
for v in r3 both; do
  MACVO_HIP_LIB=$P/libmacvo_hip_r4_$v.so timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --exact-steps 0 --config4-steps 0 --no-decoder-leg 2>&1 | tail -1 > gpurun_out/r04_ab_bench_$v.json
  python - >> $L <<PY
import json
d=json.load(open("gpurun_out/r04_ab_bench_$v.json")); r=d["roofline"]
print("$v bench", d["value"], "fps", d["ms_per_step"], "ms | gemm", r["avg_launch_us"], "us alone", r.get("isolated_avg_launch_us"), "| timeline", d.get("timeline"))
PY
done
timeout 900 python -m pytest tests/test_gpu_bench.py tests/test_gpu_split.py tests/test_gpu_plugins.py -x -q 2>&1 | tail -6 >> $L
cat $L
