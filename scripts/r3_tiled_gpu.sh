#!/bin/bash
cd $GRAFT_REPO_ROOT
L=gpurun_out/r3_tiled.log; : > $L
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_lanes.py -x -q -k "tiled" 2>&1 | tail -6 >> $L
python - >> $L 2>&1 <<'PY'
import sys; sys.path.insert(0, '.')
import torch
from macvo_amd import ops
from oracle import corr
def t(fn, n=50, warm=15):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for B in (2, 6, 64):
    H, W = 60, 80; N = H * W
    vol = torch.randn(B * N, 1, H, W, device="cuda")
    g = torch.Generator().manual_seed(0)
    coords = (corr.coords_grid(B, H, W) + torch.rand(B, 2, H, W, generator=g) * 16 - 8).cuda()
    out = torch.empty(B, 81, H, W, device="cuda")
    a = t(lambda: ops.corr_lookup(vol, coords, 4, out=out))
    b = t(lambda: ops.corr_lookup(vol, coords, 4, out=out, tiled=True))
    print(f"lookup B={B}: row-major {a:.1f} us, tiled {b:.1f} us")
    del vol
PY
for T in 1 0; do echo "== config4 MV_PIPE_TILED=$T" >> $L; MV_PIPE_TILED=$T timeout 300 python bench.py --lanes 32 --steps 100 --warmup 10 --no-cpu-baseline --no-decoder-leg --exact-steps 0 --config4-steps 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'],'fps',d['ms_per_step'],'ms | GEMM',r['avg_launch_us'],'| timeline',d.get('timeline'))
" >> $L 2>&1; done
cat $L
