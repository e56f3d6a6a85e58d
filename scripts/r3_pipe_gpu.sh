cd $GRAFT_REPO_ROOT
L=gpurun_out/r3_pipe1.log; : > $L
timeout 600 python -m pytest tests/test_gpu_native.py tests/test_gpu_split.py -x -q 2>&1 | tail -5 >> $L
for w in back vol main; do echo "== pack on $w" >> $L; MV_PIPE_PACK_ON=$w timeout 300 python bench.py --steps 300 --no-cpu-baseline --config4-steps 0 --no-decoder-leg --exact-steps 0 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(d['value'], 'fps', d['ms_per_step'], 'ms/step | GEMM', r['avg_launch_us'], 'us in pipe, alone', r.get('isolated_avg_launch_us'), '| frac', r['frac'], r['kernel'])" >> $L 2>&1; done
timeout 600 python bench.py --steps 300 2>&1 | tail -1 > gpurun_out/r3_bench_default.json
python -c "
import json; d = json.load(open('gpurun_out/r3_bench_default.json'))
print(json.dumps({k: d[k] for k in ('value','ms_per_step','roofline','exact_fp32','config4','parity')}, indent=1)[:3500])" >> $L 2>&1
cat $L
