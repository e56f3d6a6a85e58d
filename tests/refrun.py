#!/usr/bin/env python
"""The reference's OWN, UNMODIFIED caller — ``Odometry/MACVO.py`` (``from_config`` :84-92, ``initialize`` :158-171,
``run_pair`` :173-337, ``terminate`` :373-378) driven by the reference's own ``IOdometry.receive_frames``
(``Odometry/Interface.py:25-70``, which also writes ``poses.npy`` / ``tensor_map.npz``) — executed twice on the same frames:

  ``--mode ref``   every module is the reference's class, on the CPU (``device: cpu``; the TwoFrame_PGO optimizer in its
                   sequential mode — its ``parallel`` mode spawns a child interpreter, which would not see the import shims);
  ``--mode hip``   the SAME config with ONLY the ``type:`` strings of the hot-path modules swapped to the ``HIP_*`` plugins of
                   ``mac-vo_amd/plugins.py`` (and ``device: cuda``); everything else — MACVO.run_pair itself, VisualMap, the motion
                   model, the keyframe selector, the outlier filters, MotionInterpolate — stays the reference's code.

This is test infrastructure (VERDICT r3 "What's missing" #2, "Next round" #1).  The reference's Python tree is imported from
``$MACVO_REFERENCE_ROOT``, else ``/root/reference`` (build container), else ``oracle/_ref/pyref`` (byte-compiled by
``oracle/build_ref.py``, git-ignored, travels to the GPU box like the built ``.so``).  Third-party packages the path never calls
(cv2, jaxtyping, rerun, ...) are the inert placeholders of ``tests/golden/make_golden.py``; PyPose is ``tests/golden/pypose_shim.py``.

The learned network is absent (weights + FlowFormer submodule): its place is taken by a REPLAY of network outputs —
``model.inference(A, B) -> (flow, cov)`` returns the stored maps of the frame whose index is encoded in pixel (0, 0) of ``A[0]``.
Two frontend wirings are exercised:
  * ``replay``   ``Replay_FlowFormerCovFrontend`` = the reference's ``FlowFormerCovFrontend`` (Module/Frontend/Frontend.py:143-262) with only
                 ``__init__`` replaced (no checkpoint to load); ``estimate_pair`` / ``inference_2_depth`` / ``inference_2_match`` are the
                 reference's.  HIP side: ``HIP_FlowFormerCovFrontend`` with ``config.model`` = the same replay object.
  * ``gtcov``    the reference's weight-free ``FrontendCompose(ApplyGTDepthCov(.), ApplyGTMatchCov(.))`` (Frontend.py:131-160,
                 StereoDepth.py:236-266, Matching.py:281-309) around ``FixtureDepth`` / ``FixtureMatcher`` = ground truth of the unit-test
                 asset plus a small deterministic error (with the bare ``GTDepth`` / ``GTMatcher`` inside, the error — hence every
                 covariance — is exactly 0 and the covariance-aware selector keeps no point: ``q < 1.5 * median = 0`` never holds).

    python tests/refrun.py --mode ref --case tartan_fast --out /tmp/ref.npz
    python tests/refrun.py --golden            # build container: rewrites tests/golden/macvo_run.npz (tartanair cases, ref mode)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path
from types import SimpleNamespace as NS

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PYREF = os.path.join(ROOT, "oracle", "_ref", "pyref")

CASES = {
    # mirrors Scripts/UnitTest/assets/test_config/MACVO/MACVO.yaml (CovAwareSelector, FilterCompose of the three filters, icp)
    "tartan_icp": dict(seq="tartanair", frontend="replay", selector="CovAwareSelector", graph="icp", outlier="compose", mapping=False),
    # mirrors Config/Experiment/MACVO/MACVO_Fast.yaml (NoDepth selector, CovarianceSanityFilter, disp graph, dense mapping)
    "tartan_fast": dict(seq="tartanair", frontend="replay", selector="CovAwareSelector_NoDepth", graph="disp", outlier="sanity", mapping=True),
    "tartan_reproj": dict(seq="tartanair", frontend="replay", selector="CovAwareSelector", graph="reproj", outlier="sanity", mapping=True),
    # the reference's weight-free frontends
    "tartan_gtcov": dict(seq="tartanair", frontend="gtcov", selector="CovAwareSelector", graph="icp", outlier="compose", mapping=True),
    # the benchmark's workload (640x480 synthetic stream, tests/synth.make_sequence) through the reference's loop
    "synth_fast": dict(seq="synthetic", frontend="replay", selector="CovAwareSelector_NoDepth", graph="disp", outlier="sanity", mapping=True),
}
GOLDEN_CASES = ("tartan_icp", "tartan_fast", "tartan_reproj", "tartan_gtcov")     # host-independent inputs (exact fp32 arithmetic only)


def reference_root() -> str | None:
    for p in (os.environ.get("MACVO_REFERENCE_ROOT"), "/root/reference", PYREF):
        if p and os.path.isdir(os.path.join(p, "Odometry")):
            return p
    return None


# ------------------------------------------------------------------------------------------------------------ frames
def _tex(H, W, a, b, c, m):
    """integer texture k(u, v) in [0, m): exact on every host"""
    vs, us = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    return (us * a + vs * b + (us * vs) % c) % m


def tartanair_maps():
    """The reference's unit-test asset as NETWORK OUTPUTS.  Everything is built with exact fp32 arithmetic (+ - * / on small integers,
    no exp / randn) so the same bits come out on every host: flow[0,0] = -fx*b/depth (stereo sample; the frontend takes |.|),
    flow[1] = ground-truth optical flow t-1 -> t, cov = 2^-3 * (1 + k/16) with an integer texture k (isolated minima for the 7x7 NMS),
    x 64 where the flow is flagged invalid or the depth is sky."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "tartanair_p000.npz"))
    depth = torch.from_numpy(z["depth"])
    flow = (torch.from_numpy(z["flow_u16"].astype(np.float32)) - 32768.0) / 64.0
    fmask = torch.from_numpy(z["flow_mask"])
    poses = torch.from_numpy(z["poses"])
    fx, fy, cx, cy = [float(v) for v in z["K"]]
    bl = float(z["baseline"])
    n, H, W = depth.shape
    cam = dict(fx=fx, fy=fy, cx=cx, cy=cy, baseline=bl, H=H, W=W)
    k = [_tex(H, W, 7, 13, 11, 17).float(), _tex(H, W, 5, 3, 13, 19).float(), _tex(H, W, 11, 5, 7, 23).float(), _tex(H, W, 3, 17, 5, 13).float()]
    frames = []
    for t in range(n):
        fl = torch.zeros(2, 2, H, W)
        fl[0, 0] = -(fx * bl) / depth[t]
        cov = torch.stack([torch.stack([0.125 * (1 + k[0] / 16), 0.125 * (1 + k[1] / 16)]),
                           torch.stack([0.125 * (1 + k[2] / 16), 0.125 * (1 + k[3] / 16)])])
        cov[0] = cov[0] * (1.0 + 63.0 * (depth[t] > 100.0).float())
        if t > 0:
            fl[1] = flow[t - 1]
            bad = (fmask[t - 1] != 0) | (depth[t - 1] > 100.0)
            cov[1] = cov[1] * (1.0 + 63.0 * bad.float())
        frames.append(dict(flow=fl, cov=cov, gt_depth=depth[t][None, None],
                           gt_flow=(flow[t] if t < n - 1 else torch.zeros(2, H, W))[None]))
    return cam, frames, poses


def synthetic_maps(n_frames: int, seed: int = 0):
    """tests/synth.make_sequence (the benchmark's stream): cov = exp(2 * log-sigma) evaluated ONCE on the CPU so that both modes replay
    the same bits (torch.randn / exp are not bit-reproducible across hosts: these cases are only compared live, ref vs hip)."""
    sys.path.insert(0, ROOT)
    from tests import synth

    cam, frs, poses = synth.make_sequence(n_frames=n_frames, C=8, iters=1, seed=seed, pool=1)
    return cam, [dict(flow=f["flow"], cov=torch.exp(2.0 * f["logcov"])) for f in frs], torch.stack(poses)


class ReplayNet:
    """Stands in for ``FlowFormerCov`` (flownet.py:35-44): ``inference(A, B) -> (flow [B,2,H,W], exp(2 * log-sigma) [B,2,H,W])``.
    Sample 0 = stereo pair of the frame in ``A[0]``, sample 1 = temporal pair (previous frame -> that frame)."""

    def __init__(self, frames, device):
        self.maps = [(f["flow"].to(device), f["cov"].to(device)) for f in frames]

    def inference(self, A, B):
        t = int(round(float(A[0, 0, 0, 0]) * 255.0))
        flow, cov = self.maps[t]
        return flow[: A.shape[0]].clone(), cov[: A.shape[0]].clone()

    def eval(self):
        return self


def make_stereo_frames(ref, cam, frames, poses):
    pp = sys.modules["pypose"]
    H, W = cam["H"], cam["W"]
    K = torch.tensor([[cam["fx"], 0, cam["cx"]], [0, cam["fy"], cam["cy"]], [0, 0, 1.0]])[None]
    img = (_tex(H, W, 3, 5, 7, 251).float() / 255.0)[None, None].repeat(1, 3, 1, 1)
    out = []
    for t, f in enumerate(frames):
        L = img.clone()
        L[0, :, 0, 0] = t / 255.0                                            # frame index for ReplayNet
        st = ref.StereoData(T_BS=pp.identity_SE3(1), K=K.clone(), baseline=torch.tensor([cam["baseline"]]), time_ns=[1_000_000 * (t + 1)],
                            height=H, width=W, imageL=L, imageR=img.clone(), gt_flow=f.get("gt_flow"), flow_mask=None,
                            gt_depth=f.get("gt_depth"))
        out.append(ref.StereoFrame(idx=[t], time_ns=[1_000_000 * (t + 1)], gt_pose=pp.SE3(poses[t].float()[None]), stereo=st))
    return out


# ------------------------------------------------------------------------------------------------------------ reference import
_REF = None


def import_reference():
    global _REF
    if _REF is not None:
        return _REF
    root = reference_root()
    assert root is not None, "needs the reference's Python tree: /root/reference, $MACVO_REFERENCE_ROOT or oracle/_ref/pyref (python oracle/build_ref.py)"
    os.environ["MACVO_REFERENCE_ROOT"] = root
    sys.path.insert(0, ROOT)
    from tests.golden import make_golden as MG

    MG.import_reference()
    if not torch.cuda.is_available():
        # TwoFramePGO/Optimizer.py:83 evaluates torch.cuda.current_stream() although the (inactive) Timer ignores it
        torch.cuda.current_stream = lambda *a, **k: None
    import Module
    import Odometry.MACVO as OM
    from DataLoader import StereoFrame
    from DataLoader.Interface import StereoData
    from Module.Frontend import Frontend as FE
    from Module.Frontend.Matching import IMatcher
    from Module.Frontend.StereoDepth import IStereoDepth
    from Utility.Sandbox import Sandbox

    class Replay_FlowFormerCovFrontend(FE.FlowFormerCovFrontend):
        """The reference's class with only the constructor replaced (no FlowFormer submodule / checkpoint here)."""

        def __init__(self, config):
            FE.IFrontend.__init__(self, config)
            self.model = config.model

    class FixtureDepth(IStereoDepth):
        """ground-truth depth * (1 + e), |e| <= 1/64, e != 0 (exact fp32 arithmetic); disparity from the estimate"""

        @property
        def provide_cov(self): return False

        def estimate(self, frame):
            dev = self.config.device
            H, W = frame.height, frame.width
            sgn = ((_tex(H, W, 3, 5, 7, 2) * 2 - 1)).float()
            e = (sgn * (1.0 + _tex(H, W, 7, 13, 11, 16).float()) / 1024.0)[None, None]
            z = (frame.gt_depth * (1.0 + e)).to(dev)
            dcov = (0.0625 * (1.0 + _tex(H, W, 5, 3, 13, 19).float() / 16.0))[None, None].to(dev)
            return IStereoDepth.Output(depth=z, disparity=(frame.frame_baseline * frame.fx) / z, disparity_uncertainty=dcov)

        @classmethod
        def is_valid_config(cls, config): return

    class FixtureMatcher(IMatcher):
        """ground-truth flow + e, 1/32 <= |e| <= 1/2 px (exact fp32 arithmetic)"""

        @property
        def provide_cov(self): return False

        def forward(self, frame_t1, frame_t2):
            H, W = frame_t1.height, frame_t1.width
            su = ((_tex(H, W, 3, 5, 7, 2) * 2 - 1)).float()
            sv = ((_tex(H, W, 5, 7, 3, 2) * 2 - 1)).float()
            e = torch.stack([su * (1.0 + _tex(H, W, 11, 5, 7, 16).float()) / 32.0, sv * (1.0 + _tex(H, W, 3, 17, 5, 16).float()) / 32.0])[None]
            return IMatcher.Output(flow=(frame_t1.gt_flow + e).to(self.config.device))

        @classmethod
        def is_valid_config(cls, config): return

    _REF = NS(Module=Module, OM=OM, StereoFrame=StereoFrame, StereoData=StereoData, Sandbox=Sandbox, root=root)
    return _REF


# ------------------------------------------------------------------------------------------------------------ config
def make_config(case: dict, mode: str):
    hip = mode == "hip"
    dev = "cuda" if hip else "cpu"
    P = "HIP_" if hip else ""
    sel = case["selector"]
    kp_args = NS(device=dev, kernel_size=7, mask_width=32, max_match_cov=100.0)
    if sel == "CovAwareSelector":
        kp_args = NS(device=dev, kernel_size=7, mask_width=32, max_depth="auto", max_depth_cov=250.0, max_match_cov=100.0)
    if case["frontend"] == "replay":
        fe = NS(type=("HIP_FlowFormerCovFrontend" if hip else "Replay_FlowFormerCovFrontend"),
                args=NS(weight="", device=dev, enc_dtype="fp32", dec_dtype="fp32", decoder_depth=12, enforce_positive_disparity=False))
    else:
        fe = NS(type="FrontendCompose", args=NS(
            depth=NS(type="ApplyGTDepthCov", args=NS(module=NS(type="FixtureDepth", args=NS(device=dev)))),
            match=NS(type="ApplyGTMatchCov", args=NS(module=NS(type="FixtureMatcher", args=NS(device=dev))))))
    if case["outlier"] == "compose":
        outlier = NS(type="FilterCompose", args=NS(filter_args=[
            NS(type="CovarianceSanityFilter", args=NS()), NS(type="SimpleDepthFilter", args=NS(min_depth=0.05, max_depth="auto")),
            NS(type="LikelyFrontOfCamFilter", args=NS())]))
    else:
        outlier = NS(type="CovarianceSanityFilter", args=NS())
    od = NS(
        name="refrun",
        args=NS(device=dev, edgewidth=32, num_point=200, match_cov_default=0.25, profile=False, mapping=case["mapping"]),
        cov=NS(obs=NS(type=P + "MatchCovariance", args=NS(device=dev, kernel_size=31, match_cov_default=0.25, min_depth_cov=0.05, min_flow_cov=0.25))),
        keypoint=NS(type=P + sel, args=kp_args),
        mappoint=NS(type=P + "MappingPointSelector", args=NS(max_depth=5.0, max_depth_cov=0.005, mask_width=32)),
        frontend=fe,
        motion=NS(type="StaticMotionModel", args=NS()),
        outlier=outlier,
        postprocess=NS(type="MotionInterpolate", args=NS()),
        keyframe=NS(type="AllKeyframe", args=NS()),
        # reference: CPU, sequential (see module docstring).  HIP: the solve is a GPU kernel on its own stream (parallel: true)
        optimizer=NS(type=P + "TwoFrame_PGO", args=NS(device=dev, vectorize=True, parallel=hip, graph_type=case["graph"], autodiff=False)),
    )
    return NS(Odometry=od)


# ------------------------------------------------------------------------------------------------------------ run
def maps_from_file(path: str):
    """network outputs handed over by another process (bench.py: the benchmark's own frames): flow, cov [n,2,2,H,W]; cam = JSON"""
    z = np.load(path)
    cam = json.loads(str(z["cam"]))
    flow, cov = torch.from_numpy(z["flow"]), torch.from_numpy(z["cov"])
    poses = torch.from_numpy(z["poses"]) if "poses" in z.files else torch.tensor([[0, 0, 0, 0, 0, 0, 1.0]]).repeat(flow.shape[0], 1).double()
    return cam, [dict(flow=flow[t], cov=cov[t]) for t in range(flow.shape[0])], poses


def run_case(name: str, mode: str, n_frames: int | None = None, seed: int = 1234, quiet: bool = True, maps_file: str = "",
             mapping: int = -1, threads: int = 0, graph: str = "", repeat_maps: int = 1, cprofile: bool = False) -> dict:
    case = dict(CASES[name])
    if graph:
        case["graph"] = graph
    if mapping >= 0:
        case["mapping"] = bool(mapping)
    if threads > 0:
        torch.set_num_threads(threads)
    ref = import_reference()
    if mode == "hip":
        import macvo_amd.interfaces as I
        import macvo_amd.plugins  # noqa: F401  registers the HIP_* classes in the reference's own registries
        assert I.USING_REFERENCE
    if maps_file:
        cam, maps, poses = maps_from_file(maps_file)
    elif case["seq"] == "tartanair":
        cam, maps, poses = tartanair_maps()
    else:
        cam, maps, poses = synthetic_maps(n_frames or 8)
    if n_frames:
        maps, poses = maps[:n_frames], poses[:n_frames]
    if repeat_maps > 1:      # a longer stream for steady-state timing: the same network outputs again (the loop does not care that the scene repeats)
        maps, poses = maps * repeat_maps, poses.repeat(repeat_maps, 1)
    frames = make_stereo_frames(ref, cam, maps, poses)
    cfg = make_config(case, mode)
    ref.OM.MACVO.is_valid_config(cfg.Odometry)                     # the reference's own validator, before anything is added to the config
    if case["frontend"] == "replay":
        cfg.Odometry.frontend.args.model = ReplayNet(maps, "cuda" if mode == "hip" else "cpu")
    if quiet:
        from Utility.PrettyPrint import GlobalConsole
        GlobalConsole.quiet = True
    torch.manual_seed(seed)                                        # the selectors consume the global CPU generator (KeypointSelector.py:331,404)
    system = ref.OM.MACVO.from_config(cfg)
    stamps = [time.perf_counter()]

    def on_frame(frame, sysm, pb):
        if mode == "hip":
            torch.cuda.synchronize()
        stamps.append(time.perf_counter())

    prof = None
    if cprofile:
        import cProfile

        prof = cProfile.Profile()
    with tempfile.TemporaryDirectory() as tmp:
        box = ref.Sandbox(Path(tmp))
        if prof is not None:
            prof.enable()
        system.receive_frames(frames, box, on_frame_finished=on_frame)
        if prof is not None:
            prof.disable()
        assert system.terminated and os.path.exists(box.path("tensor_map.npz")), \
            "receive_frames swallowed an exception (Odometry/Interface.py:62-70): see the log above"
        tm = dict(np.load(box.path("tensor_map.npz")))
        out = {f"map/{k}": v for k, v in tm.items()}
        out["poses_npy"] = np.load(box.path("poses.npy"))
    if prof is not None:
        import pstats

        st = pstats.Stats(prof)
        skip = ("marshal", "builtins.exec", "importlib", "_imp.", "open_code", "posix.stat", "builtins.compile", "_io.")    # one-time imports, not per-frame costs
        rows = sorted(((v[2], v[0], k) for k, v in st.stats.items() if not any(w in k[2] or w in k[0] for w in skip)), reverse=True)[:3]   # (tottime, calls, (file, line, name))
        out["host_top"] = np.array(json.dumps([{"fn": f"{os.path.basename(k[0])}:{k[1]}:{k[2]}", "ms_per_frame": round(tt * 1e3 / max(1, len(frames) - 1), 3),
                                                "calls_per_frame": round(nc / max(1, len(frames) - 1), 1)} for tt, nc, k in rows]))
    out["frame_s"] = np.diff(np.array(stamps))
    out["gt_poses"] = poses.numpy()
    out["n_frames"] = np.array(len(frames))
    out["classes"] = np.array(json.dumps({k: type(getattr(system, k)).__name__ for k in
                                          ("Frontend", "KeypointSelector", "MappointSelector", "ObsCovModel", "Optimizer", "OutlierFilter")}))
    return out


TOL_KEYS = {"map/match//obs1_covTc": 5e-5, "map/match//obs2_covTc": 5e-5, "map/points//cov_Tw": 5e-5, "map/points//pos_Tw": 1e-5}


def compare_runs(a: dict, b: dict, pose_tol=1e-4) -> list[str]:
    """Differences between two runs; ``map/*`` = the keys of the reference's ``VisualMap.serialize`` as its ``receive_frames`` wrote them
    to ``tensor_map.npz``.  Bit-exact: every integer / bool store and edge table, the keypoints (``pixel1_uv``), their tracked positions, every
    gathered per-pixel value, colours, camera rows.  Float tolerances: 3x3 covariances 5e-5 of the matrix scale (the HIP covariance model's
    parity bar), 3-D points 1e-5 relative, poses ``pose_tol`` per component (north_star: 1e-4 m / 1e-4 rad)."""
    bad = []
    keys = sorted(k for k in a if k.startswith("map/"))
    if keys != sorted(k for k in b if k.startswith("map/")):
        return [f"key sets differ: {set(keys) ^ set(k for k in b if k.startswith('map/'))}"]
    for k in keys:
        x, y = a[k], b[k]
        if x.shape != y.shape:
            bad.append(f"{k}: shape {x.shape} vs {y.shape}")
        elif x.size == 0:
            continue
        elif k == "map/frames//pose":
            d = np.abs(x.astype(np.float64) - y.astype(np.float64)).max()
            if d > pose_tol:
                bad.append(f"{k}: max |d| = {d:.3e} > {pose_tol}")
        elif k in TOL_KEYS:
            xf, yf = x.astype(np.float64), y.astype(np.float64)
            scale = np.abs(xf).reshape(xf.shape[0], -1).max(axis=1).reshape((-1,) + (1,) * (xf.ndim - 1))   # per row (per 3x3 matrix / point)
            err = (np.abs(xf - yf) / np.maximum(scale, 1e-30)).max()
            if not err <= TOL_KEYS[k]:
                bad.append(f"{k}: max row-relative diff {err:.3e} > {TOL_KEYS[k]}")
        elif not np.array_equal(x, y, equal_nan=(x.dtype.kind == "f")):
            bad.append(f"{k}: not bit-equal ({int((x != y).sum())} of {x.size} entries differ)")
    d = np.abs(a["poses_npy"][:, 1:] - b["poses_npy"][:, 1:]).max()
    if d > pose_tol or not np.array_equal(a["poses_npy"][:, 0], b["poses_npy"][:, 0]):
        bad.append(f"poses.npy: max |d| = {d:.3e}")
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["ref", "hip"], default="ref")
    ap.add_argument("--case", default="tartan_fast")
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--out", default="")
    ap.add_argument("--golden", action="store_true")
    ap.add_argument("--loud", action="store_true")
    ap.add_argument("--maps-file", default="", help=".npz with flow / cov [n,2,2,H,W] and a JSON `cam`: replay these network outputs")
    ap.add_argument("--mapping", type=int, default=-1, help="override the case's dense-mapping flag (0 / 1)")
    ap.add_argument("--graph", default="", choices=["", "icp", "reproj", "disp"], help="override the case's residual graph")
    ap.add_argument("--threads", type=int, default=0, help="torch.set_num_threads for the run (0 = torch's default)")
    ap.add_argument("--repeat-maps", type=int, default=1, help="replay the network outputs this many times in a row (steady-state timing)")
    ap.add_argument("--cprofile", action="store_true", help="run the loop under cProfile and report the three largest host costs (own time) per frame")
    a = ap.parse_args()
    if a.golden:
        assert os.path.isdir("/root/reference/Odometry"), "--golden runs in the build container (needs /root/reference)"
        out = {}
        for name in GOLDEN_CASES:
            r = run_case(name, "ref", quiet=not a.loud)
            for k, v in r.items():
                if k != "frame_s":
                    out[f"{name}/{k}"] = v
            print(name, "frames/s", 1.0 / r["frame_s"][1:].mean(), "matches", r["map/match//pixel1_uv"].shape[0])
        path = os.path.join(ROOT, "tests", "golden", "macvo_run.npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path) / 1e6, "MB")
        return
    r = run_case(a.case, a.mode, n_frames=a.frames or None, quiet=not a.loud, maps_file=a.maps_file, mapping=a.mapping, threads=a.threads, graph=a.graph,
                 repeat_maps=a.repeat_maps, cprofile=a.cprofile)
    if a.out:
        np.savez_compressed(a.out, **r)
    print(json.dumps({"case": a.case, "mode": a.mode, "frames": int(r["n_frames"]), "s_per_frame": float(r["frame_s"][1:].mean()),
                      "s_per_frame_steady": float(np.sort(r["frame_s"][2:])[: max(1, (len(r["frame_s"]) - 2) * 3 // 4)].mean()) if len(r["frame_s"]) > 3 else None,
                      "s_per_frame_after5": float(r["frame_s"][5:].mean()) if len(r["frame_s"]) > 6 else None,
                      "s_per_frame_median": float(np.median(r["frame_s"][1:])),
                      "host_top": json.loads(str(r["host_top"])) if "host_top" in r else None,
                      "threads": torch.get_num_threads(),
                      "matches": int(r["map/match//pixel1_uv"].shape[0]), "classes": json.loads(str(r["classes"]))}))


if __name__ == "__main__":
    main()
