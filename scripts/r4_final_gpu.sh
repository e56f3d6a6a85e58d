#!/bin/bash
# Round-4 closing evidence: GPU suite, smoke, kernel alone-times, the bench lines (default incl. the end_to_end leg / the driver's 20-step form),
# rocprofv3 kernel stats of the bench and of the end-to-end tool (what the learned frontend's 43 ms are made of), PGO stamps.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/r04b_suite.log; : > $L
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 >> $L
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3 >> $L
timeout 300 python tools/kernel_bench.py pgo volume_split lookup --iters 50 2>&1 | grep -v amdgpu.ids > gpurun_out/r04b_kernels_kernel_bench.log
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/r04b_bench_default_line.json
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --end-to-end-frames 0 2>&1 | tail -1 > gpurun_out/r04b_bench_steps20_line.json
bash scripts/profile_gpu.sh r04b_bench --config4-steps 0 --no-decoder-leg --exact-steps 0 >> $L 2>&1
grep '^{"metric' gpurun_out/prof_r04b_bench/bench.log | tail -1 > gpurun_out/r04b_bench_profiled_line.json
find gpurun_out/prof_r04b_bench -name "*kernel_trace.csv" -delete      # (gpurun_out travels back only below 64 MiB: the stats are what is kept)
OUT=gpurun_out/prof_r04b_e2e; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT" -o trace -- python "$OLDPWD/tools/end_to_end.py" --frames 14 --warmup 4 --variants hooked ) > $OUT/e2e.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r04b_e2e_kernel_stats.csv
grep '^{"end_to_end' $OUT/e2e.log | tail -1 > gpurun_out/r04b_e2e_profiled_line.json
rm -rf $OUT
timeout 120 python profiles/probes/pgo_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04b_pgo_stamps.log
python - >> $L 2>&1 <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04b_bench_*_line.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d["roofline"]
    print(f.split("/")[-1], d["value"], "fps", d["ms_per_step"], "ms |", r["kernel"], r["avg_launch_us"], "us frac", r["frac"], "alone", r.get("isolated_avg_launch_us"), "| timeline", d.get("timeline"))
    if d.get("parity"):
        p = d["parity"]; print("    parity kp", p["keypoints_bit_exact_frames"], "/", p["frames"], p["max_pose_dt_m"], "| vol/lookup", p.get("volume_and_lookups"), "| vs ref", {k: v for k, v in (p.get("vs_reference_loop") or {}).items() if k != "what"})
        print("    cpu_baseline", d["cpu_baseline"])
    if d.get("end_to_end"):
        e = d["end_to_end"]; print("    end_to_end", {v: {k: x for k, x in e[v].items() if k not in ("hooks", "classes")} for v in ("hooked", "unhooked") if v in e}, e.get("hooked_over_unhooked"))
PY
cat $L; cat gpurun_out/r04b_kernels_kernel_bench.log | grep -E "^pgo|^volume|^lookup" | head -30; tail -4 gpurun_out/r04b_pgo_stamps.log
