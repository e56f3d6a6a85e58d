"""The reference's REAL in-tree RAFTFlowCovNet (Module/Network/PWCNet/RAFTCov.py:45-107) on the GPU with its five 81-channel
local correlations per frame running in mv_local_corr81 — through plugins.HIP_TartanVOCovMatcher, instantiated by the
reference's own registry.  Needs the reference's Python tree (Module/, Utility/): it is NOT part of
this repository; scripts/raft_gpu.sh ships it inside the gpurun command line and points MACVO_REFERENCE_ROOT at the unpacked copy.

Checks: (1) every FunctionCorrelation call of the network reaches the HIP kernel and equals the oracle's definition of the op on
the same tensors; (2) the matcher's records have the reference's shapes / mask quirk; (3) the whole network on the GPU (HIP
correlation, MIOpen convolutions) agrees with the same network (same random weights) on the CPU with the oracle's correlation."""
import os, sys, time, warnings
warnings.filterwarnings("ignore")
os.environ.setdefault("MIOPEN_LOG_LEVEL", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from types import SimpleNamespace as NS
import torch
from tests.golden import make_golden as MG
sys.meta_path.insert(0, MG._Finder())
MG.pypose_shim.install()
torch.compile = lambda f=None, **kw: f
import Module
import macvo_amd.interfaces as I
import macvo_amd.plugins as P
from macvo_amd import ops
from oracle import corr as ocorr
assert I.USING_REFERENCE
dev = torch.device("cuda:0")
calls = []
def hip_corr(tenFirst, tenSecond):
    out = P.FunctionCorrelation(tenFirst, tenSecond)
    ref = ocorr.local_corr81(tenFirst.float().cpu(), tenSecond.float().cpu())
    calls.append((tuple(tenFirst.shape), float((out.cpu() - ref).abs().max()), float(ref.abs().max())))
    return out
P.HIP_TartanVOCovMatcher.correlation = staticmethod(hip_corr)
torch.manual_seed(0)
m = Module.IMatcher.instantiate("HIP_TartanVOCovMatcher", NS(weight="", device="cuda"))
H, W = 448, 640
g = torch.Generator().manual_seed(1)
fa = NS(imageL=torch.rand(1, 3, H, W, generator=g).to(dev), height=H, width=W)
fb = NS(imageL=torch.rand(1, 3, H, W, generator=g).to(dev), height=H, width=W)
with torch.no_grad():
    out = m.estimate(fa, fb)
torch.cuda.synchronize()
print("kernel of the 5 FunctionCorrelation call sites:", [c[0] for c in calls])
for shp, err, mag in calls:
    print(f"   {shp}: max |HIP - oracle| = {err:.3e} (|ref| up to {mag:.3e})")
    assert err <= 1e-5 * max(1.0, mag)
assert len(calls) == 5 and out.flow.shape == (1, 2, H, W) and out.cov.shape == (1, 3, H, W) and not out.mask.any()
assert torch.isfinite(out.flow).all() and (out.cov[:, :2] > 0).all()
# the same network on the CPU with the oracle's correlation
P.HIP_TartanVOCovMatcher.correlation = staticmethod(lambda tenFirst, tenSecond: ocorr.local_corr81(tenFirst.float(), tenSecond.float()))
P.patch_reference_correlation(P.HIP_TartanVOCovMatcher.correlation)
cuda_orig = torch.Tensor.cuda
torch.Tensor.cuda = lambda self, *a, **k: self          # the reference's warp() hard-codes .cuda() (pwc_model.py)
cpu_net = m.model.to("cpu")
cpu_net.device = "cpu"                                   # RAFTFlowCovNet.inference moves its inputs to self.device (RAFTCov.py:156)
with torch.no_grad():
    fl_c, cov_c = cpu_net.inference(fa.imageL.cpu(), fb.imageL.cpu())
torch.Tensor.cuda = cuda_orig
df = (out.flow.cpu() - fl_c).abs().max().item()
dc = ((out.cov.cpu()[:, :2] - cov_c).abs() / cov_c.abs().clamp_min(1e-6)).max().item()
print(f"GPU (HIP correlation + MIOpen) vs CPU (oracle correlation), same weights: max |dflow| = {df:.3e} px (|flow| up to {fl_c.abs().max():.2f}), max rel dcov = {dc:.3e}")
assert df <= 5e-2 and dc <= 5e-2
# timing of the plugin's estimate on the GPU
P.HIP_TartanVOCovMatcher.correlation = None
P.patch_reference_correlation(None)
m.model.to(dev)
m.model.device = "cuda"
with torch.no_grad():
    for _ in range(3): m.estimate(fa, fb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m.estimate(fa, fb)
    torch.cuda.synchronize()
print(f"HIP_TartanVOCovMatcher.estimate on MI355X: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per 640x448 pair (random weights, fp32, 12 GRU iterations)")
print("RAFT-GPU OK")
