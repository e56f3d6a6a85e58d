cd $GRAFT_REPO_ROOT
L=gpurun_out/r3_pipe2.log; : > $L
timeout 600 python -m pytest tests/test_gpu_native.py -x -q 2>&1 | tail -5 >> $L
timeout 900 python bench.py --steps 300 2>&1 | tail -1 > gpurun_out/r3_bench_default.json
python -c "
import json; d = json.load(open('gpurun_out/r3_bench_default.json'))
def rl(r): return None if r is None else {k: r[k] for k in ('kernel','avg_launch_us','frac','isolated_avg_launch_us','isolated_frac') if k in r}
print('default', d['value'], 'fps', d['ms_per_step'], rl(d['roofline']))
for k, v in (d.get('other_precisions') or {}).items(): print(k, v['value'], v['ms_per_step'], rl(v['roofline']))
c = d['config4']; print('config4', c['value'], c['ms_per_step'], rl(c['roofline']))
print('parity', d['parity']['keypoints_bit_exact_frames'], d['parity']['max_pose_dt_m'], d['rte_vs_oracle'], 'cpu', d['cpu_baseline']['value'])
print('decoder', d['decoder_loop'])" >> $L 2>&1
python tools/lane_timeline.py 1 300 2>&1 | tail -5 >> $L
TL_PRECISION=f16x2 python tools/lane_timeline.py 1 300 2>&1 | tail -5 >> $L
cat $L
