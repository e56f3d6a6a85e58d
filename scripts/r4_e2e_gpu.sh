#!/bin/bash
# FlowFormerCov-shaped host network on the MI355X: hooked vs unhooked outputs (exploration of the tolerances), then the end-to-end tool.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/r04_e2e.log; : > $L
timeout 600 python - >> $L 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import flowformer_host as fh
from macvo_amd import plugins
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(2)
a, b = torch.rand(2, 3, 480, 640, generator=g).to(dev), torch.rand(2, 3, 480, 640, generator=g).to(dev)
def build(enc, dec, depth, hooked, skip_pe=False):
    torch.manual_seed(7)
    m = fh.FlowFormerCovHost(fh.demo_cfg(decoder_depth=depth), enc, dec).to(dev).eval()
    names = []
    if hooked:
        if skip_pe:
            pe = m.memory_encoder.cost_perceiver_encoder.patch_embed
            proj = pe.proj; pe.proj = torch.nn.Identity()
            names = plugins.install_flowformer_hooks(m); pe.proj = proj
        else:
            names = plugins.install_flowformer_hooks(m)
    return m, names
for enc, dec, depth in ((torch.float32, torch.float32, 3), (torch.float16, torch.bfloat16, 12)):
    with torch.inference_mode():
        m0, _ = build(enc, dec, depth, False)
        f0, c0 = m0.inference(a, b); del m0
        for skip in (True, False):
            m1, names = build(enc, dec, depth, True, skip)
            f1, c1 = m1.inference(a, b)
            print(enc, dec, depth, "skip_pe" if skip else "all hooks", names)
            print("   flow |max| %.4f  max diff %.3e  mean diff %.3e | cov ratio max dev %.3e" % (f0.abs().max().item(), (f1 - f0).abs().max().item(), (f1 - f0).abs().mean().item(), (c1 / c0 - 1).abs().max().item()))
            torch.cuda.synchronize(); t = time.time()
            for _ in range(3): m1.inference(a, b)
            torch.cuda.synchronize(); print("   hooked ms per B=2 inference", (time.time() - t) / 3 * 1e3)
            del m1
        m0, _ = build(enc, dec, depth, False)
        m0.inference(a, b); torch.cuda.synchronize(); t = time.time()
        for _ in range(3): m0.inference(a, b)
        torch.cuda.synchronize(); print("   unhooked ms per B=2 inference", (time.time() - t) / 3 * 1e3)
        del m0
PY
timeout 900 python tools/end_to_end.py --frames 20 --warmup 4 2>&1 | tail -3 >> $L
cat $L
