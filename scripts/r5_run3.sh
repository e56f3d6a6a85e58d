#!/bin/bash
# round 5, call 3: new bench legs + tests, upsample 16-bit variants
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_flowformer_host.py tests/test_gpu_patch_embed.py tests/test_gpu_bench.py -q -m gpu -x -k "not end_to_end" 2>&1 | tail -8
for cfg in "1 4" "2 2" "2 4" "1 2"; do set -- $cfg; MV_UPS_PX=$1 MV_UPS_NSX=$2 timeout 120 python profiles/probes/r5_upsample_ab.py 2>&1 | grep "NSX" | sed "s/^/PX=$1 /"; done
timeout 900 python bench.py --steps 20 > gpurun_out/r05a_bench_steps20_line.json 2> gpurun_out/r05a_bench_steps20.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05a_bench_steps20_line.json') if l.startswith('{')][-1])
print("value", d["value"], "ms/step", d["ms_per_step"])
print("roofline", {k:d["roofline"][k] for k in ("avg_launch_us","frac","isolated_avg_launch_us")})
print("timeline", d["timeline"])
print("kernels", json.dumps(d["kernels"], indent=1))
print("plugin_path", json.dumps(d["plugin_path"], indent=1))
print("patch_embed", json.dumps(d["patch_embed"], indent=1)[:1500])
print("decoder_loop", d["decoder_loop"])
print("parity", {k:v for k,v in d["parity"].items() if k!="volume_and_lookups"}, d["parity"]["volume_and_lookups"])
PY
tail -5 gpurun_out/r05a_bench_steps20.err
