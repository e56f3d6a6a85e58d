#!/bin/bash
cd $GRAFT_REPO_ROOT
L=gpurun_out/r3_pgo2.log; : > $L
for S in 0 1; do echo "== MV_PGO_SPEC=$S" >> $L; MV_PGO_SPEC=$S python tools/kernel_bench.py pgo --iters 100 2>&1 | grep "pgo" | head -4 >> $L; done
MV_PGO_SPEC=2 python - >> $L 2>&1 <<'PY'
import sys; sys.path.insert(0, '.')
import torch
from macvo_amd import ops
from oracle import pgo
from tests.test_gpu_backend import _to_batch
for seed in (6, 1, 2, 3):
    prob, _ = pgo.make_synthetic_problem(n=200, seed=seed)
    b = _to_batch([prob], torch.device("cuda"))
    pose, info = ops.pgo_solve(b, "disp")
    torch.cuda.synchronize()
    print("seed", seed, "steps", info[0, 1].item(), "rejects(last step)", info[0, 2].item(), "rounds*1000+loop iterations", info[0, 3].item())
PY
cat $L
