#!/usr/bin/env python
"""Generate golden vectors by RUNNING THE REAL REFERENCE MODULES (build container only: needs /root/reference).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

What is executed from ``/root/reference`` (unmodified, imported in place):
  * Module/KeypointSelector.py   CovAwareSelector, CovAwareSelector_NoDepth, MappingPointSelector
  * Module/Covariance/Project2to3.py   MatchCovariance.estimate, Covariance_2to3_full
  * Utility/Math.py   gaussain_full_kernels (eager: torch.compile is replaced by identity so results do not depend
                      on inductor code generation)
  * Module/Frontend/StereoDepth.py   disparity_to_depth, disparity_to_depth_cov;  Frontend.py retrieve_pixels;
    Utility/Point.py filterPointsInRange
  * Module/Optimization/TwoFramePGO/{Graphs,Optimizer}.py + PyposeOptimizers.py   TwoFrame_PGO._optimize with the
    analytic graphs (icp / reproj / disp) — on top of tests/golden/pypose_shim.py (PyPose itself is not installable)

Third-party packages the reference imports at module scope but that the hot path never calls (cv2, jaxtyping,
typeguard, yacs, torchvision, evo, rerun, ...) are satisfied by inert placeholder modules.
The GPU box has no /root/reference: tests only read the .npz files this script writes.
"""
from __future__ import annotations

import hashlib
import importlib.abc
import importlib.machinery
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MACVO_REFERENCE_ROOT", "/root/reference")   # (tools/raft_gpu_probe.py unpacks the .py tree elsewhere on the GPU box)
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from tests import synth  # noqa: E402
from tests.golden import pypose_shim  # noqa: E402

MISSING = {"cv2", "yacs", "evo", "rerun", "torchvision", "timm", "cupy", "xformers", "flow_vis", "h5py", "wandb", "onnx",
           "tensorrt", "jaxtyping", "typeguard", "onnxruntime", "pycuda", "kornia", "numba", "open3d"}


class _Dummy:
    def __init__(self, *a, **k): pass
    def __call__(self, *a, **k): return _Dummy()
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Dummy()
    def __getitem__(self, k): return _Dummy()
    def __class_getitem__(cls, k): return cls
    def __iter__(self): return iter(())


class _Sub:
    def __getitem__(self, k): return object


class _Anything(types.ModuleType):
    __path__: list = []

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return type(k, (_Dummy,), {})


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in MISSING:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        m = _Anything(spec.name)
        if spec.name == "jaxtyping":
            for k in ("Float32", "Bool", "Float", "Float64", "Int", "Int64", "UInt8", "Shaped"):
                setattr(m, k, _Sub())
            m.jaxtyped = lambda typechecker=None: (lambda c: c)
        if spec.name == "typeguard":
            m.typechecked = lambda f: f
        m.__version__ = "999.0"   # version gates at import time (e.g. Utility/Visualize/Rerun_Visualize.py:21)
        return m

    def exec_module(self, m):
        pass


def import_reference():
    sys.meta_path.insert(0, _Finder())
    pypose_shim.install()
    torch.compile = lambda f=None, **kw: f  # OnCallCompiler -> eager
    import Module.KeypointSelector as KS
    import Module.Covariance.Project2to3 as P23
    import Module.Frontend.StereoDepth as SD
    import Module.Frontend.Frontend as FE
    import Module.Optimization.TwoFramePGO.Optimizer as OPT
    import Module.Optimization.TwoFramePGO.Graphs as GR
    import Utility.Math as UM
    import Utility.Point as UP
    return SimpleNamespace(KS=KS, P23=P23, SD=SD, FE=FE, OPT=OPT, GR=GR, UM=UM, UP=UP)


def sha(*tensors) -> str:
    h = hashlib.sha256()
    for t in tensors:
        h.update(t.contiguous().numpy().tobytes())
    return h.hexdigest()


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# ------------------------------------------------------------------------------------------------------------
def gen_selectors(ref):
    cases = {}
    frame = SimpleNamespace(fx=320.0, frame_baseline=0.25, height=0, width=0)
    # (name, H, W, seeds, nan_frac, store_inputs)
    for name, H, W, nan_frac, store in (("small", 96, 136, 0.02, True), ("full", 480, 640, 0.003, False)):
        fc = synth.flow_cov_maps(H, W, seed=2, nan_frac=nan_frac)
        d0, d0c = synth.depth_maps(H, W, 3)
        d1, d1c = synth.depth_maps(H, W, 4)
        match = SimpleNamespace(cov=fc.clone(), mask=None)
        dep0 = SimpleNamespace(depth=d0, cov=d0c, mask=None)
        dep1 = SimpleNamespace(depth=d1, cov=d1c, mask=None)

        cfg = SimpleNamespace(device="cpu", mask_width=16 if name == "small" else 32, kernel_size=7, max_match_cov=100.0)
        sel = ref.KS.CovAwareSelector_NoDepth(cfg)
        torch.manual_seed(1234)
        px_nd = sel.select_point(frame, 200, dep0, dep1, match)

        cfg2 = SimpleNamespace(device="cpu", mask_width=cfg.mask_width, max_depth="auto", kernel_size=7, max_depth_cov=250.0,
                               max_match_cov=100.0)
        sel2 = ref.KS.CovAwareSelector(cfg2)
        torch.manual_seed(4321)
        px_full = sel2.select_point(frame, 200, dep0, dep1, SimpleNamespace(cov=fc.clone(), mask=None))

        cfg3 = SimpleNamespace(max_depth=20.0, max_depth_cov=0.2, mask_width=cfg.mask_width)
        sel3 = ref.KS.MappingPointSelector(cfg3)
        torch.manual_seed(7)
        px_map = sel3.select_point(frame, 300, dep0, dep1, None)

        cases[f"{name}_nodepth_px"] = px_nd
        cases[f"{name}_full_px"] = px_full
        cases[f"{name}_mapping_px"] = px_map
        cases[f"{name}_meta"] = np.array([H, W, cfg.mask_width])
        cases[f"{name}_sha"] = np.frombuffer(bytes.fromhex(sha(fc, d0, d0c, d1, d1c)), dtype=np.uint8)
        if store:
            cases.update({f"{name}_fc": fc, f"{name}_d0": d0, f"{name}_d0c": d0c, f"{name}_d1": d1, f"{name}_d1c": d1c})
    save("selector", **cases)


def gen_covariance(ref):
    H, W, n = 120, 160, 48
    depth, dcov = synth.depth_maps(H, W, 3)
    kp = synth.keypoints(n, H, W, 5, border=20)
    g = torch.Generator().manual_seed(8)
    kpf = kp.float() + torch.rand(n, 2, generator=g)
    fcov = torch.exp(2 * 0.5 * torch.randn(n, 3, generator=g))
    fcov[:, 2] = 0.2 * torch.randn(n, generator=g) * fcov[:, :2].min(dim=1).values
    fcov[:4, 0] = 0.01
    fcov[:4, 2] = 0.0
    frame = SimpleNamespace(fx=160.0, fy=150.0, cx=80.0, cy=60.0)
    cfg = SimpleNamespace(kernel_size=31, match_cov_default=0.25, min_flow_cov=0.25, min_depth_cov=0.05, device="cpu")
    model = ref.P23.MatchCovariance(cfg)
    dest = SimpleNamespace(depth=depth, cov=dcov)
    out = {"depth": depth, "dcov": dcov, "kp_int": kp, "kp_float": kpf, "flow_cov_in": fcov.clone(),
           "K": np.array([frame.fx, frame.fy, frame.cx, frame.cy])}
    fc1 = fcov.clone()
    out["cov_int_flowcov"] = model.estimate(frame, kp, dest, None, fc1)           # kp1-style call (flow cov given)
    out["flow_cov_after"] = fc1                                                      # in-place clamp
    fc2 = fcov.clone()
    out["cov_float_flowcov"] = model.estimate(frame, kpf, dest, None, fc2)
    dc = dcov[0, 0, kp[:, 1], kp[:, 0]].contiguous()
    out["depth_cov_kp"] = dc
    out["cov_int_nodefault"] = model.estimate(frame, kp, dest, dc, None)            # flow_cov None -> given depth cov
    s0 = torch.ones(n, 3) * 0.25
    s0[:, 2] = 0
    out["cov_int_default_sigma"] = model.estimate(frame, kp, dest, dc, s0)          # kp0-style call (MACVO.py:241)
    # building blocks
    cm = ref.P23.create_2x2_matrix([[fc1[:, 0], fc1[:, 2]], [fc1[:, 2], fc1[:, 1]]], n, torch.device("cpu"))
    out["gauss_kernels"] = ref.UM.gaussain_full_kernels(cm, 31)[:6]
    save("covariance", **out)


def gen_frontend_bits(ref):
    g = torch.Generator().manual_seed(21)
    H, W = 64, 96
    disp = torch.rand(1, 1, H, W, generator=g) * 40 + 0.5
    dcov = torch.exp(torch.randn(1, 1, H, W, generator=g))
    bl, fx = 0.25, 320.0
    depth = ref.SD.disparity_to_depth(disp, bl, fx)
    depth_cov = ref.SD.disparity_to_depth_cov(disp, dcov, bl, fx)
    uv = torch.stack([torch.rand(50, generator=g) * (W - 1), torch.rand(50, generator=g) * (H - 1)], 1)
    flow = torch.randn(1, 2, H, W, generator=g)
    got = ref.FE.IFrontend.retrieve_pixels(uv, flow)
    inr = ref.UP.filterPointsInRange(uv, (8, W - 8), (8, H - 8))
    save("frontend_bits", disp=disp, dcov=dcov, depth=depth, depth_cov=depth_cov, uv=uv, flow=flow, retrieved=got,
         in_range=inr, blfx=np.array([bl, fx]))


def gen_pgo(ref):
    """Run the reference's TwoFrame_PGO._optimize (analytic graphs) on seeded problems from oracle.pgo.make_synthetic_problem."""
    from oracle import pgo as opgo

    # Optimizer.py:83 evaluates torch.cuda.current_stream() even though the (inactive) Timer ignores it; there is
    # no GPU in the build container, so hand it a placeholder
    torch.cuda.current_stream = lambda *a, **k: None

    # cases 4-6 are chosen to ENTER the reject loop mid-solve (bad prior: 5 rejections in the first LM step of
    # reproj / disp; heavy outliers), cases 0-3 mostly reject only at convergence (16 rejections in the last step)
    cases = [dict(n=200, seed=6), dict(n=200, seed=7, outlier_frac=0.1), dict(n=37, seed=8),
             dict(n=120, seed=10, trans_sigma=0.4, rot_sigma=0.08),
             dict(n=200, seed=11, trans_sigma=1.0, rot_sigma=0.3, outlier_frac=0.3),
             dict(n=60, seed=12, trans_sigma=2.0, rot_sigma=0.5),
             dict(n=200, seed=13, outlier_frac=0.5)]
    out = {}
    SOP = pypose_shim.StopOnPlateau
    # both readings of StopOnPlateau's reject rule (see pypose_shim.StopOnPlateau): "" = threshold 1 (default),
    # "_r16" = threshold = optimizer.reject
    for tag, thr in (("", 1), ("_r16", 16)):
        SOP.STOP_ON_REJECT = thr
        for gi, gname in enumerate(("icp", "reproj", "disp")):
            cfg = SimpleNamespace(graph_type=gname, device="cpu", vectorize=True, parallel=False, autodiff=False)
            ctx = ref.OPT.TwoFrame_PGO.init_context(cfg)
            for ci, c in enumerate(cases):
                prob, _ = opgo.make_synthetic_problem(**c)
                obs = SimpleNamespace(data={
                    "pixel2_uv": prob.pixel2_uv, "pixel2_d": prob.pixel2_d, "pixel2_disp": prob.pixel2_disp,
                    "pixel2_disp_cov": prob.pixel2_disp_cov, "pixel2_uv_cov": prob.pixel2_uv_cov, "obs2_covTc": prob.obs2_covTc})
                pts = SimpleNamespace(data={"pos_Tw": prob.pos_Tw, "cov_Tw": prob.cov_Tw})
                n = prob.pos_Tw.shape[0]
                gin = ref.GR.GraphInput(frame_idx=torch.tensor([1]), from_idx=torch.tensor([0]),
                                        init_motion=pypose_shim.SE3(prob.init_pose.reshape(1, 7).clone()),
                                        baseline=torch.tensor([prob.baseline], dtype=torch.float32), observations=obs, points=pts,
                                        images_intrinsic=prob.K, edges_index=torch.zeros(n, dtype=torch.long), device="cpu")
                _, gout = ref.OPT.TwoFrame_PGO._optimize(ctx, gin)
                pose = gout.motion.detach().as_subclass(torch.Tensor).reshape(7).double()
                sch = SOP.last_instance
                out[f"{gname}_{ci}_pose{tag}"] = pose
                # {outer LM steps, reject_count of the last step, final robust loss, max reject_count over the steps}
                out[f"{gname}_{ci}_stats{tag}"] = np.array([sch.steps, sch.optimizer.reject_count, float(sch.optimizer.loss),
                                                            max(sch.reject_hist)], dtype=np.float64)
                out[f"{gname}_{ci}_case"] = np.array([c["n"], c["seed"], c.get("outlier_frac", 0.0), c.get("trans_sigma", 0.1), c.get("rot_sigma", 0.02)])
    SOP.STOP_ON_REJECT = 1
    save("pgo", **out)


def gen_visual_map(ref):
    """Drive the REAL map classes (Module/Map/VisualMap.py, Graph.py, Template.py) with the call sequence of MACVO.initialize
    (Odometry/MACVO.py:162-169) and MACVO.run_pair (:244-311, 339-347) on the synthetic per-frame tables of
    tests/synth.map_sequence, then VisualMap.serialize (:104-116), the poses.npy rows of Odometry/Interface.py:47-51 and the
    REAL MotionInterpolate.elaborate_map (Module/MapProcessor.py:57-76; its PyPose calls run on the shim)."""
    import Module.Map as MM
    from Module.MapProcessor import MotionInterpolate
    pp = sys.modules["pypose"]

    meta, frames = synth.map_sequence()
    vmap = MM.VisualMap()
    min_num_point = 10

    def push_keyframe(fr, est_pose):
        return vmap.frames.push(MM.FrameNode.init({
            "pose": est_pose, "T_BS": meta["T_BS"].reshape(1, 7), "need_interp": torch.tensor([False], dtype=torch.bool),
            "time_ns": torch.tensor([fr["time_ns"]], dtype=torch.long), "K": meta["K"].reshape(1, 3, 3),
            "baseline": torch.tensor([meta["baseline"]])}))

    inputs = {}
    prev_idx = int(push_keyframe(frames[0], torch.tensor([[0, 0, 0, 0, 0, 0, 1.0]])).item())     # MACVO.initialize
    for t, fr in enumerate(frames[1:], start=1):
        n, mask = fr["n"], fr["valid"]
        v = fr["vals"]
        match_obs = MM.MatchObs.init({                                                       # MACVO.py:244-266
            "pixel1_uv": fr["kp0"], "pixel2_uv": fr["kp1"], "pixel1_d": v[0].unsqueeze(-1), "pixel2_d": v[4].unsqueeze(-1),
            "pixel1_disp": v[1].unsqueeze(-1), "pixel2_disp": v[5].unsqueeze(-1), "pixel1_disp_cov": v[2].unsqueeze(-1),
            "pixel2_disp_cov": v[6].unsqueeze(-1), "pixel1_d_cov": v[3].unsqueeze(-1), "pixel2_d_cov": v[7].unsqueeze(-1),
            "pixel1_uv_cov": fr["sigma0"], "pixel2_uv_cov": fr["sigma1"], "obs1_covTc": fr["cov0"], "obs2_covTc": fr["cov1"]})
        match_obs = match_obs[mask]                                                          # :270
        num_match_orig = len(vmap.match)
        point_idx = vmap.points.push(MM.PointNode.init({"pos_Tw": fr["pos_Tw"], "cov_Tw": fr["cov0w"], "color": fr["color"]})[mask])
        frame_idx = push_keyframe(fr, fr["prior"].reshape(1, 7))                             # :282
        prev_frame_idx = torch.tensor([prev_idx], dtype=torch.long)
        match_idx = vmap.match.push(match_obs)
        num_match_kp = len(match_obs)
        vmap.point2match.add(point_idx, match_idx)                                           # :288-293
        vmap.match2point.set(match_idx, point_idx)
        vmap.frame2match.add(prev_frame_idx, torch.tensor([num_match_orig]), torch.tensor([num_match_kp]))
        vmap.frame2match.add(frame_idx, torch.tensor([num_match_orig]), torch.tensor([num_match_kp]))
        vmap.match2frame1.set(match_idx, torch.empty((num_match_kp,), dtype=torch.long).fill_(prev_frame_idx.item()))
        vmap.match2frame2.set(match_idx, torch.empty((num_match_kp,), dtype=torch.long).fill_(frame_idx.item()))
        prev_idx = int(frame_idx.item())
        if match_idx.size(0) < min_num_point:                                               # :303-307 lost track: no mapping either
            vmap.frames.data["need_interp"][frame_idx] = True
        else:                                                                               # dense-mapping tail, :313-337
            num_map_orig = len(vmap.map_points)
            vmap.map_points.push(MM.PointNode.init({"pos_Tw": fr["map_pos_Tw"], "cov_Tw": fr["map_cov"], "color": fr["map_color"]}))
            vmap.frame2map.add(frame_idx, torch.tensor([num_map_orig], dtype=torch.long),
                               torch.tensor([fr["map_pos_Tw"].size(0)], dtype=torch.long))
        vmap.frames.data["pose"][frame_idx] = fr["opt"].reshape(1, 7)                        # write_graph_data (Optimizer.py:104-108)
        for k in ("valid", "kp0", "kp1", "vals", "sigma0", "sigma1", "cov0", "cov1", "pos_Tw", "cov0w", "color", "prior", "opt",
                  "map_pos_Tw", "map_cov", "map_color"):
            inputs[f"in/{t}/{k}"] = fr[k]
        inputs[f"in/{t}/time_ns"] = np.array(fr["time_ns"], dtype=np.int64)
    out = {f"ser/{k}": np.array(v, copy=True) for k, v in vmap.serialize().items()}   # serialize() returns views of the live stores
    # VisualMap.serialize leaves the map-point store itself out (VisualMap.py:104-116): record it next to the edges that index it
    nmp = len(vmap.map_points)
    for k in ("pos_Tw", "cov_Tw", "color"):
        out[f"mp/{k}"] = vmap.map_points.data[k].tensor[:nmp].clone().numpy()
    out["mp/project_last"] = vmap.frame2map.project(torch.tensor([prev_idx], dtype=torch.long)).numpy()   # get_frame2map of the newest frame
    sensor_poses = pp.SE3(vmap.frames.data["pose"].tensor)                                    # Interface.py:47-51
    T_BS = pp.SE3(vmap.frames.data["T_BS"].tensor)
    body = (T_BS @ sensor_poses @ T_BS.Inv()).tensor().cpu().numpy()
    time_ns = vmap.frames.data["time_ns"].tensor.cpu().numpy()[:, np.newaxis]
    out["poses_npy"] = np.concatenate([time_ns, body], axis=-1)
    assert np.array_equal(out["ser/frames//pose"], vmap.frames.data["pose"].tensor.numpy())
    frames_store, interp_idx = MotionInterpolate(None).elaborate_map(vmap.frames)
    out["interp/pose"] = frames_store.data["pose"].tensor.clone().numpy()
    out["interp/idx"] = interp_idx.numpy()
    out.update(inputs)
    out["meta/K"], out["meta/T_BS"], out["meta/baseline"] = meta["K"], meta["T_BS"], np.array(meta["baseline"], dtype=np.float32)
    out["meta/n_frames"] = np.array(len(frames))
    out["meta/time0"] = np.array(frames[0]["time_ns"], dtype=np.int64)
    save("visual_map", **out)


def gen_filters():
    """The real observation filters (Module/OutlierFilter.py:91-145) on seeded rows with NaN / inf covariances, depths around
    the gates and variances around the 2-sigma front-of-camera test (one variant carries the -1 "no covariance" placeholder)."""
    import Module.OutlierFilter as OF

    n = 96
    g = torch.Generator().manual_seed(11)
    cov1 = torch.eye(3, dtype=torch.float64).repeat(n, 1, 1) * (0.1 + torch.rand(n, 1, 1, generator=g, dtype=torch.float64))
    cov2 = cov1.clone() * 2
    cov1[3, 0, 1] = float("nan")
    cov2[7, 2, 2] = float("inf")
    cov1[11, 1, 1] = float("-inf")
    cov2[12, 0, 0] = float("nan")
    d1 = torch.rand(n, 1, generator=g) * 12
    d2 = torch.rand(n, 1, generator=g) * 12
    d1[20], d2[21], d1[22], d2[23] = 0.05, 0.05, 10.0, 10.0                       # exactly on the gates (strict compares)
    c1 = (torch.rand(n, 1, generator=g) * 3) ** 2
    c2 = (torch.rand(n, 1, generator=g) * 3) ** 2
    c1[30] = (d1[30] / 2) ** 2                                                     # d - 2*sqrt(c) == 0 -> rejected (strict >)
    c2[31] = float("nan")
    out = dict(cov1=cov1, cov2=cov2, d1=d1, d2=d2, c1=c1, c2=c2)
    for tag, cc1 in (("", c1), ("_placeholder", torch.where(torch.arange(n)[:, None] == 40, torch.tensor(-1.0), c1))):
        class _Bundle:          # the two things the filters use of a TensorBundle: .data[...] and len()
            data = {"obs1_covTc": cov1, "obs2_covTc": cov2, "pixel1_d": d1, "pixel2_d": d2, "pixel1_d_cov": cc1,
                    "pixel2_d_cov": c2}

            def __len__(self):
                return n

        vals = _Bundle()
        dev = torch.device("cpu")
        out["sanity" + tag] = OF.CovarianceSanityFilter(SimpleNamespace()).filter(vals, dev)
        out["depth" + tag] = OF.SimpleDepthFilter(SimpleNamespace(min_depth=0.05, max_depth=10.0)).filter(vals, dev)
        out["front" + tag] = OF.LikelyFrontOfCamFilter(SimpleNamespace()).filter(vals, dev)
    save("filters", **out)


def gen_upsample():
    """The in-tree twin of FlowFormer's convex upsampling: GaussianGRU.upsample_flow (Module/Network/PWCNet/pwc_cov/gru.py:40-52),
    executed unmodified (module loaded by path; its only sibling import is the torch-only attention.py)."""
    import importlib.util
    import types

    base = os.path.join(REF, "Module/Network/PWCNet/pwc_cov") + "/"
    pkg = types.ModuleType("pwc_cov_pkg")
    pkg.__path__ = [base]
    sys.modules["pwc_cov_pkg"] = pkg
    mods = {}
    for name in ("attention", "gru"):
        spec = importlib.util.spec_from_file_location(f"pwc_cov_pkg.{name}", base + name + ".py")
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"pwc_cov_pkg.{name}"] = m
        spec.loader.exec_module(m)
        mods[name] = m
    out = {}
    for ci, (n, h, w, seed) in enumerate([(2, 6, 7, 0), (1, 8, 9, 1), (3, 1, 1, 2)]):
        g = torch.Generator().manual_seed(seed)
        flow = torch.randn(n, 2, h, w, generator=g) * 4
        mask = torch.randn(n, 576, h, w, generator=g) * 2
        out[f"c{ci}_flow"], out[f"c{ci}_mask"] = flow, mask
        out[f"c{ci}_out"] = mods["gru"].GaussianGRU.upsample_flow(SimpleNamespace(kernel_size=3), flow, mask)
    save("upsample", **out)


if __name__ == "__main__":
    assert os.path.isdir(REF), "make_golden.py must run where /root/reference exists"
    ref = import_reference()
    only = set(sys.argv[1:])          # e.g. `make_golden.py visual_map`: regenerate one file
    gens = {"selector": lambda: gen_selectors(ref), "covariance": lambda: gen_covariance(ref), "frontend_bits": lambda: gen_frontend_bits(ref),
            "pgo": lambda: gen_pgo(ref), "visual_map": lambda: gen_visual_map(ref), "upsample": gen_upsample, "filters": gen_filters}
    for name, fn in gens.items():
        if not only or name in only:
            fn()
