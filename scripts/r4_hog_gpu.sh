#!/bin/bash
# A/B: GEMM workgroups that claim the whole LDS of their CU (MV_SPLIT_LDS_KB) on fewer CUs (MV_SPLIT_WGS): LDS-using small kernels get the rest.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/r04_hog.log; : > $L
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --exact-steps 0 --config4-steps 0 --no-decoder-leg 2>&1 | tail -1 > gpurun_out/r04_hog_$tag.json
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --exact-steps 0 --config4-steps 0 --no-decoder-leg 2>&1 | tail -1 > gpurun_out/r04_hog20_$tag.json
  python - >> $L <<PY
import json
d=json.load(open("gpurun_out/r04_hog_$tag.json")); r=d["roofline"]; e=json.load(open("gpurun_out/r04_hog20_$tag.json"))
print("$tag", d["value"], "fps | 20-step", e["value"], "| gemm", r["avg_launch_us"], "us alone", r.get("isolated_avg_launch_us"), "| timeline", d.get("timeline"))
PY
}
run base A=1
run w224hog MV_SPLIT_WGS=224 MV_SPLIT_LDS_KB=160
run w192hog MV_SPLIT_WGS=192 MV_SPLIT_LDS_KB=160
run w240hog MV_SPLIT_WGS=240 MV_SPLIT_LDS_KB=160
run w224 MV_SPLIT_WGS=224
run w256hog MV_SPLIT_LDS_KB=160
run base2 A=1
cat $L
