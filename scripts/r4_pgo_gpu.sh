#!/bin/bash
# (history: ran at commit 5f0b... when the round-3 kernel was still selectable with MV_PGO_V1=1; that knob is gone)
# Round-4 PGO kernel (explicit FMAs + lean build): parity tests, kernel == host twin, per-phase stamps, alone-times and the bench line against the
# round-3 kernel (MV_PGO_V1=1).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/r04_pgo.log; : > $L
timeout 300 python -m pytest tests/test_gpu_backend.py tests/test_gpu_golden.py -k "pgo" -x -q 2>&1 | tail -5 >> $L
for V in 0 1; do
  echo "== MV_PGO_V1=$V" >> $L
  MV_PGO_V1=$V timeout 120 python tools/kernel_bench.py pgo --iters 100 2>&1 | grep "^pgo" >> $L
done
echo "== MV_PGO_SPEC=0 (new kernel, sequential reject loop)" >> $L
MV_PGO_SPEC=0 timeout 120 python tools/kernel_bench.py pgo --iters 100 2>&1 | grep "^pgo" | head -2 >> $L
echo "== stamps (new kernel)" >> $L
timeout 120 python profiles/probes/pgo_stamps.py 2>&1 | grep -v amdgpu.ids >> $L
timeout 400 python -m pytest tests/test_gpu_pipeline.py tests/test_macvo_run.py tests/test_gpu_native.py -x -q 2>&1 | tail -5 >> $L
for rep in 1 2; do
for V in 1 0; do
  MV_PGO_V1=$V timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --exact-steps 0 --config4-steps 0 --no-decoder-leg 2>&1 | tail -1 > gpurun_out/r04_pgo_bench_v1_$V.json
  python - >> $L <<PY
import json
d=json.load(open("gpurun_out/r04_pgo_bench_v1_$V.json")); r=d["roofline"]
print("MV_PGO_V1=$V bench", d["value"], "fps", d["ms_per_step"], "ms | gemm", r["avg_launch_us"], "us | timeline", d.get("timeline"), "| rte_vs_oracle", d.get("rte_vs_oracle"))
PY
  MV_PGO_V1=$V timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --exact-steps 0 --config4-steps 0 --no-decoder-leg 2>&1 | tail -1 > gpurun_out/r04_pgo_bench20_v1_$V.json
  python - >> $L <<PY
import json
d=json.load(open("gpurun_out/r04_pgo_bench20_v1_$V.json"))
print("MV_PGO_V1=$V 20-step line", d["value"], "fps", d["ms_per_step"], "ms")
PY
done
done
cat $L
