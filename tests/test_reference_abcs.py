"""The `USING_REFERENCE` branch of mac-vo_amd/interfaces.py: inside a MAC-VO checkout the HIP plugins must subclass the
reference's OWN ABCs (so `SubclassRegistry.instantiate` finds them by name with no change to Odometry/MACVO.py,
Utility/Extensions/SubclassRegistry.py:24-48) and the reference's own `MACVO.is_valid_config` (Odometry/MACVO.py:137-156)
must accept `Config/Experiment/MACVO/MACVO_Fast.yaml` with ONLY its `type:` strings swapped.

Runs in a fresh interpreter (the mirror classes of this repository and the reference's must not mix in one process) with the
import shims of tests/golden/make_golden.py standing in for third-party packages the hot path never calls (cv2, jaxtyping,
rerun, ...; PyPose = tests/golden/pypose_shim.py).  Needs /root/reference: skipped on the GPU box."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

SCRIPT = r'''
import sys
from pathlib import Path
from types import SimpleNamespace as NS
sys.path.insert(0, %(root)r)
from tests.golden import make_golden as MG
MG.import_reference()                      # shims + sys.path for the reference checkout
import Module                               # the reference's plugin package (Module/__init__.py:1-11)
import macvo_amd.interfaces as I
assert I.USING_REFERENCE, "the real-ABC branch must be taken when the reference is importable"
import macvo_amd.plugins as P

# 1. the plugins ARE subclasses of the reference's own interfaces and sit in their registries under cls.name()
pairs = [(Module.IKeypointSelector, "HIP_CovAwareSelector_NoDepth"), (Module.IKeypointSelector, "HIP_CovAwareSelector"),
         (Module.IKeypointSelector, "HIP_MappingPointSelector"), (Module.ICovariance2to3, "HIP_MatchCovariance"),
         (Module.IOptimizer, "HIP_TwoFrame_PGO"), (Module.IFrontend, "HIP_FlowFormerCovFrontend"),
         (Module.IFrontend, "HIP_CUDAGraph_FlowFormerCovFrontend"), (Module.IStereoDepth, "HIP_FlowFormerCovDepth"),
         (Module.IMatcher, "HIP_FlowFormerCovMatcher")]
for iface, name in pairs:
    cls = iface.get_class(name)
    assert cls is getattr(P, name) and issubclass(cls, iface), (iface, name)
assert I.IStereoDepth is Module.IStereoDepth and I.IMatcher.Output is Module.IMatcher.Output
from Module.Optimization.TwoFramePGO.Graphs import GraphInput
assert I.GraphInput is GraphInput

# 2. a name clash with a reference class is rejected by the reference's own registry (SubclassRegistry.py:42-46)
try:
    class CovAwareSelector_NoDepth(Module.IKeypointSelector):   # noqa: F811 - same name as the reference's class
        def select_point(self, *a): ...
        @classmethod
        def is_valid_config(cls, config): ...
    raise SystemExit("duplicate name was accepted")
except NameError:
    pass

# 3. the reference's own loader + validator on MACVO_Fast.yaml with only the `type:` strings swapped
from Utility.Config import load_config
import Odometry.MACVO as OM
cfg, _ = load_config(Path(%(ref)r) / "Config/Experiment/MACVO/MACVO_Fast.yaml")
od = cfg.Odometry
OM.MACVO.is_valid_config(od)                                     # the untouched config is valid to begin with
swap = {"MatchCovariance": "HIP_MatchCovariance", "CovAwareSelector_NoDepth": "HIP_CovAwareSelector_NoDepth",
        "MappingPointSelector": "HIP_MappingPointSelector",
        "CUDAGraph_FlowFormerCovFrontend": "HIP_CUDAGraph_FlowFormerCovFrontend", "TwoFrame_PGO": "HIP_TwoFrame_PGO"}
for sec in (od.cov.obs, od.keypoint, od.mappoint, od.frontend, od.optimizer):
    sec.type = swap[sec.type]
od.optimizer.args.device = "cuda"                                # the HIP solver runs on the GPU (the reference's is a CPU child)
OM.MACVO.is_valid_config(od)
# exact-key-set rule still applies to the plugins (Utility/Extensions/Testable.py:37-41)
od.keypoint.args.bogus = 1
try:
    OM.MACVO.is_valid_config(od)
    raise SystemExit("excess key was accepted")
except (KeyError, AssertionError):
    pass
del od.keypoint.args.bogus

# 3b. all three shipped experiment configs validate with only the type strings swapped — Paper_Reproduce.yaml carries `autodiff: true`
#     (:103-109): the HIP solver accepts it (analytic Jacobians = the reference's verified equivalent) instead of refusing the file
import warnings
swap["CovAwareSelector"] = "HIP_CovAwareSelector"
for name in ("MACVO_Performant.yaml", "Paper_Reproduce.yaml"):
    c2, _ = load_config(Path(%(ref)r) / "Config/Experiment/MACVO" / name)
    o2 = c2.Odometry
    for sec in (o2.cov.obs, o2.keypoint, o2.mappoint, o2.frontend, o2.optimizer):
        sec.type = swap[sec.type]
    OM.MACVO.is_valid_config(o2)
    if o2.optimizer.args.autodiff:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            c = Module.IOptimizer.get_class(o2.optimizer.type).init_context(NS(**{**vars(o2.optimizer.args), "parallel": False}))
        assert c["graph_type"] == "icp" and any("analytic" in str(x.message) for x in w)

# 4. instantiate through the reference's registry exactly as MACVO.from_config does (Odometry/MACVO.py:84-92) — the
#    classes whose constructors need neither a GPU nor network weights
kp = Module.IKeypointSelector.instantiate(od.keypoint.type, od.keypoint.args)
mp = Module.IKeypointSelector.instantiate(od.mappoint.type, od.mappoint.args)
cv = Module.ICovariance2to3.instantiate(od.cov.obs.type, od.cov.obs.args)
assert isinstance(kp, P.HIP_CovAwareSelector_NoDepth) and isinstance(mp, P.HIP_MappingPointSelector)
assert isinstance(cv, P.HIP_MatchCovariance) and cv.config.kernel_size == 31
opt_cls = Module.IOptimizer.get_class(od.optimizer.type)
seq_args = NS(**{**vars(od.optimizer.args), "parallel": False})  # parallel = True creates its HIP stream: needs the GPU
ctx = opt_cls.init_context(seq_args)                             # static, picklable (Optimization/Interface.py:87-92)
assert isinstance(ctx, dict) and ctx["stream"] is None

# 5. HIP_TartanVOCovMatcher: the reference's in-tree RAFTFlowCovNet (Module/Network/PWCNet/RAFTCov.py:45-107) built by the
#    plugin with random weights, every FunctionCorrelation call site rebound — here to the oracle's CPU definition of the same
#    81-channel local correlation (the HIP kernel is checked against that definition on the GPU: test_gpu_corr::test_local_corr81)
import torch
from oracle import corr as ocorr
calls = []
def cpu_corr(tenFirst, tenSecond):
    calls.append(tuple(tenFirst.shape))
    return ocorr.local_corr81(tenFirst.float(), tenSecond.float())
assert Module.IMatcher.get_class("HIP_TartanVOCovMatcher") is P.HIP_TartanVOCovMatcher
P.HIP_TartanVOCovMatcher.correlation = staticmethod(cpu_corr)
Module.IMatcher.is_valid_config(NS(type="HIP_TartanVOCovMatcher", args=NS(weight="", device="cpu")))
torch.manual_seed(0)
torch.Tensor.cuda = lambda self, *a, **k: self      # the reference's warp() hard-codes .cuda() (pwc_model.py); no GPU in this container
m = Module.IMatcher.instantiate("HIP_TartanVOCovMatcher", NS(weight="", device="cpu"))
import Module.Network.PWCNet.RAFTCov as RC, Module.Network.PWCNet.pwc.pwc_model as PM
assert RC.FunctionCorrelation is cpu_corr and PM.FunctionCorrelation is cpu_corr
H, W = 128, 192
fa = NS(imageL=torch.rand(1, 3, H, W), height=H, width=W)
fb = NS(imageL=torch.rand(1, 3, H, W), height=H, width=W)
out = m.estimate(fa, fb)
assert len(calls) == 5 and calls[0][2:] == (H // 64, W // 64) and calls[-1][2:] == (H // 4, W // 4), calls   # five pyramid levels (pwc_model.py:178-233)
assert out.flow.shape == (1, 2, H, W) and out.cov.shape == (1, 3, H, W) and out.mask.shape == (1, 1, H, W)
assert torch.isfinite(out.flow).all() and (out.cov[:, :2] > 0).all() and (out.cov[:, 2] == 0).all()
assert not out.mask.any()        # Matching.py:260-264 as written: without padding the 0:-0 slice is empty
print("REAL-ABC OK")
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "Module")), reason="needs the reference checkout (build container only)")
def test_plugins_subclass_the_real_reference_abcs(tmp_path):
    script = tmp_path / "real_abc.py"
    script.write_text(SCRIPT % {"root": ROOT, "ref": REF})
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert out.returncode == 0 and "REAL-ABC OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
