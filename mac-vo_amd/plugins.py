"""Drop-in plugins for ``Odometry/MACVO.py``: same interfaces, same YAML ``args`` as the reference classes they
replace, every hot arithmetic step in the HIP kernels (``include/macvo_hip.h``).

    reference class (Config ``type:``)        HIP plugin
    ----------------------------------------  ----------------------------------
    CovAwareSelector_NoDepth                  HIP_CovAwareSelector_NoDepth
    CovAwareSelector                          HIP_CovAwareSelector
    MappingPointSelector                      HIP_MappingPointSelector
    MatchCovariance                           HIP_MatchCovariance
    TwoFrame_PGO                              HIP_TwoFrame_PGO
    FlowFormerCovFrontend                     HIP_FlowFormerCovFrontend  (network stays PyTorch; lookups + epilogue in HIP)
    CUDAGraph_FlowFormerCovFrontend           HIP_CUDAGraph_FlowFormerCovFrontend  (same, inference replayed as a hipGraph)
    FlowFormerCovDepth / FlowFormerCovMatcher HIP_FlowFormerCovDepth / HIP_FlowFormerCovMatcher  (for FrontendCompose configs)
    TartanVOCovMatcher                        HIP_TartanVOCovMatcher  (in-tree RAFTFlowCovNet stays PyTorch; its local correlations in HIP)
    (FlowFormerCov's volume / lookup)         install_flowformer_hooks(model)

Select them by changing only the ``type:`` strings of ``Config/Experiment/MACVO/MACVO_Fast.yaml`` (see
INTEGRATION.md); importing this module registers the classes (``SubclassRegistry`` semantics).
There is no CPU fallback: ``device`` must be a GPU device string ("cuda" is the HIP device on ROCm).
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import os

import torch

from . import ops
from .interfaces import (GraphInput, GraphOutput, ICovariance2to3, IFrontend, IKeypointSelector, IMatcher, IOptimizer,
                         IStereoDepth, _is_device)


def _num(v) -> bool:
    return isinstance(v, (int, float))


# ----------------------------------------------------------------------------------------------- selectors
class HIP_CovAwareSelector_NoDepth(IKeypointSelector):
    """``CovAwareSelector_NoDepth`` (Module/KeypointSelector.py:349-416) on the GPU; bit-exact indices."""

    def select_point(self, frame, numPoint: int, depth0_est, depth1_est, match_est) -> torch.Tensor:
        dev = torch.device(self.config.device)
        if match_est is None or match_est.cov is None:
            return _grid_select(frame, numPoint, self.config.mask_width, dev)      # GridSelector fallback (:365-366)
        fc = match_est.cov.to(dev)
        H, W = fc.shape[-2:]
        mask = None if match_est.mask is None else match_est.mask.to(dev)
        cands = ops.kp_select("nodepth", H, W, flow_cov=fc, mask_b=mask, kernel_size=self.config.kernel_size,
                              mask_width=self.config.mask_width, max_match_cov=float(self.config.max_match_cov))
        return cands.finish(numPoint)

    @classmethod
    def is_valid_config(cls, config: SimpleNamespace | None) -> None:
        cls._enforce_config_spec(config, {
            "device": _is_device,
            "mask_width": lambda m: isinstance(m, int) and m >= 0,
            "kernel_size": lambda k: isinstance(k, int) and k > 0 and (k % 2 == 1),
            "max_match_cov": lambda c: _num(c) and c > 0.0,
        })


class HIP_CovAwareSelector(IKeypointSelector):
    """``CovAwareSelector`` (Module/KeypointSelector.py:250-346)."""

    def select_point(self, frame, numPoint: int, depth0_est, depth1_est, match_est) -> torch.Tensor:
        assert depth0_est.cov is not None and depth1_est.cov is not None
        if self.config.max_depth == "auto":
            self.config.max_depth = frame.fx * frame.frame_baseline               # written back like the reference
        dev = torch.device(self.config.device)
        d0, d0c = depth0_est.depth.to(dev), depth0_est.cov.to(dev)
        d1, d1c = depth1_est.depth.to(dev), depth1_est.cov.to(dev)
        fc = match_est.cov.to(dev) if (match_est is not None and match_est.cov is not None) else None
        H, W = d0.shape[-2:]
        ma = None if depth0_est.mask is None else depth0_est.mask.to(dev)
        mb = None if (match_est is None or match_est.mask is None) else match_est.mask.to(dev)
        cands = ops.kp_select("full", H, W, flow_cov=fc, depth0=d0, depth0_cov=d0c, depth1=d1, depth1_cov=d1c, mask_a=ma,
                              mask_b=mb, kernel_size=self.config.kernel_size, mask_width=self.config.mask_width,
                              max_depth=float(self.config.max_depth), max_depth_cov=float(self.config.max_depth_cov),
                              max_match_cov=float(self.config.max_match_cov))
        return cands.finish(numPoint)

    @classmethod
    def is_valid_config(cls, config: SimpleNamespace | None) -> None:
        assert config is not None
        cls._enforce_config_spec(config, {
            "device": _is_device,
            "mask_width": lambda m: isinstance(m, int) and m >= 0,
            "max_depth": lambda d: (d == "auto") or (_num(d) and d > 0.0),
            "kernel_size": lambda k: isinstance(k, int) and k > 0 and (k % 2 == 1),
            "max_depth_cov": lambda c: _num(c) and c > 0.0,
            "max_match_cov": lambda c: _num(c) and c > 0.0,
        })


class HIP_MappingPointSelector(IKeypointSelector):
    """``MappingPointSelector`` (Module/KeypointSelector.py:78-100).  The reference's config has no ``device`` key;
    the maps' own device is used."""

    def select_point(self, frame, numPoint: int, depth0_est, depth1_est, match_est) -> torch.Tensor:
        assert depth0_est.cov is not None
        d0, d0c = depth0_est.depth, depth0_est.cov
        if not d0.is_cuda:
            d0, d0c = d0.cuda(), d0c.cuda()
        H, W = d0.shape[-2:]
        cands = ops.kp_select("mapping", H, W, depth0=d0, depth0_cov=d0c, mask_width=self.config.mask_width,
                              max_depth=float(self.config.max_depth), max_depth_cov=float(self.config.max_depth_cov))
        return cands.finish(numPoint)

    @classmethod
    def is_valid_config(cls, config: SimpleNamespace | None) -> None:
        cls._enforce_config_spec(config, {
            "max_depth": lambda v: isinstance(v, float),
            "max_depth_cov": lambda v: isinstance(v, float),
            "mask_width": lambda v: isinstance(v, int),
        })


def _grid_select(frame, numPoint: int, mask_width: int, dev) -> torch.Tensor:
    """``GridSelector.select_point`` (Module/KeypointSelector.py:222-239) — the NoDepth selector's fallback when the
    matcher gives no covariance.  Index arithmetic only (no float math), kept in torch."""
    h, w = frame.height - 2 * mask_width, frame.width - 2 * mask_width
    unit = max(1, int(math.sqrt(numPoint // 2)))
    mesh_u, mesh_v = torch.meshgrid(torch.arange(0, h, h // unit, device=dev), torch.arange(0, w, w // (unit * 2), device=dev),
                                    indexing="ij")
    return torch.stack([mesh_v.flatten(), mesh_u.flatten()], dim=1) + mask_width


# ----------------------------------------------------------------------------------------------- covariance
class HIP_MatchCovariance(ICovariance2to3):
    """``MatchCovariance`` (Module/Covariance/Project2to3.py:113-191).  Returns ``[N,3,3]`` float64 on the CPU like the
    reference (its result is pushed straight into the CPU map, Odometry/MACVO.py:265-266) and clamps the caller's
    ``flow_cov`` in place (Project2to3.py:131)."""

    def estimate(self, frame, kp, depth_est, depth_cov, flow_cov) -> torch.Tensor:
        return self.estimate_device(frame, kp, depth_est, depth_cov, flow_cov).cpu()

    def estimate_device(self, frame, kp, depth_est, depth_cov, flow_cov, rot: torch.Tensor | None = None):
        """Same as :meth:`estimate` but the result stays on the GPU (optionally also ``R cov R^T``)."""
        dev = torch.device(self.config.device)
        n = kp.size(0)
        has_flow_cov = flow_cov is not None
        if has_flow_cov:
            work = flow_cov if (flow_cov.is_cuda and flow_cov.dtype == torch.float32 and flow_cov.is_contiguous()) \
                else flow_cov.to(dev, torch.float32).contiguous()
        else:
            work = torch.full((n, 3), float(self.config.match_cov_default), dtype=torch.float32, device=dev)
            work[:, 2] = 0.0
        res = ops.match_cov(depth_est.depth.to(dev), kp.to(dev), work, None if depth_cov is None else depth_cov.to(dev),
                            frame.fx, frame.fy, frame.cx, frame.cy, kernel_size=self.config.kernel_size,
                            min_flow_cov=self.config.min_flow_cov if has_flow_cov else 0.0,   # clamp only a GIVEN flow_cov (Project2to3.py:128-135)
                            min_depth_cov=self.config.min_depth_cov,
                            use_patch_var=(has_flow_cov or depth_cov is None), rot=rot)
        if has_flow_cov and work is not flow_cov:
            flow_cov.copy_(work)                                                   # keep the in-place side effect
        return res

    @classmethod
    def is_valid_config(cls, config: SimpleNamespace | None) -> None:
        cls._enforce_config_spec(config, {
            "kernel_size": lambda k: isinstance(k, int) and k > 0 and (k % 2 == 1) and k <= 31,
            "match_cov_default": lambda c: _num(c) and c > 0.0,
            "min_flow_cov": lambda c: _num(c) and c > 0.0,
            "min_depth_cov": lambda c: _num(c) and c > 0.0,
            "device": _is_device,
        })


# ----------------------------------------------------------------------------------------------- optimizer
def _bundle(x, key):
    return x.data[key]


class HIP_TwoFrame_PGO(IOptimizer[GraphInput, dict, GraphOutput]):
    """``TwoFrame_PGO`` (Module/Optimization/TwoFramePGO/Optimizer.py:23-108) with the whole LM loop in one HIP launch.

    ``parallel: true`` in the reference means "solve in a spawned CPU process while the GPU runs the next frame's
    network" (Optimization/Interface.py:80-96).  Here the solve is a GPU kernel, so the overlap comes from a dedicated
    HIP stream: ``start_optimize`` enqueues upload + solve + download on it and returns; ``write_map`` waits on the
    completion event.  Results are identical in both modes.
    """

    _autodiff_noted = False

    def __init__(self, config: SimpleNamespace) -> None:
        self.config = config
        self.is_parallel_mode = bool(config.parallel)
        self.context = self.init_context(config)
        self.optimize_res = None
        self.has_opt_job = False
        self._pending = None

    @staticmethod
    def init_context(config) -> dict:
        if config.autodiff and not HIP_TwoFrame_PGO._autodiff_noted:
            # `autodiff: true` (Config/Experiment/MACVO/Paper_Reproduce.yaml:103-109) selects the reference's autograd Jacobians; its
            # analytic graphs are their verified equivalent (PyposeOptimizers.py:60-73 `verify_jacobian`, Graphs.py:151-231) and are
            # what the HIP solver implements — accepted, and said once
            HIP_TwoFrame_PGO._autodiff_noted = True
            import warnings
            warnings.warn("HIP_TwoFrame_PGO: autodiff: true — solving with the analytic Jacobians (the reference's verified equivalent)",
                          stacklevel=2)
        dev = torch.device("cuda" if config.device == "cpu" else config.device)   # the solve itself always runs on the GPU
        return {"graph_type": config.graph_type, "device": dev, "lm": ops.lm_default_params(),
                "stream": torch.cuda.Stream(device=dev) if config.parallel else None}

    @classmethod
    def is_valid_config(cls, config: SimpleNamespace | None) -> None:
        cls._enforce_config_spec(config, {
            "graph_type": lambda s: s in {"icp", "reproj", "disp"},
            "device": lambda v: isinstance(v, str) and (v == "cpu" or "cuda" in v),
            "vectorize": lambda b: isinstance(b, bool),
            "parallel": lambda b: isinstance(b, bool),
            "autodiff": lambda b: isinstance(b, bool),
        })

    def get_graph_data(self, global_map, frame_idx, observations=None, edges=None) -> GraphInput:
        """Same gathers as the reference (Optimizer.py:24-38) through the map's own accessors."""
        frame2opt = global_map.frames[frame_idx]
        obs = global_map.get_frame2match(frame2opt)
        pts = global_map.get_match2point(obs)
        K = frame2opt.data["K"][0]
        n = pts.data["pos_Tw"].shape[0]
        return GraphInput(frame_idx, frame_idx - 1, frame2opt.data["pose"], frame2opt.data["baseline"], obs, pts, K,
                          torch.zeros(n, dtype=torch.long), "cpu")

    @staticmethod
    def _launch(context: dict, g: GraphInput):
        dev = context["device"]
        obs, pts = g.observations, g.points
        n = _bundle(pts, "pos_Tw").shape[0]
        up = lambda t, dt: t.reshape(t.shape[0], -1).to(dev, dt, non_blocking=True).contiguous()  # noqa: E731
        K = g.images_intrinsic
        batch = ops.PGOBatch(
            offsets=torch.tensor([0, n], dtype=torch.int32).to(dev, non_blocking=True),
            init_pose=torch.as_tensor(g.init_motion).reshape(1, 7).to(dev, torch.float32, non_blocking=True),
            intrinsics=torch.stack([K[0, 0], K[1, 1], K[0, 2], K[1, 2]]).reshape(1, 4).to(dev, torch.float32, non_blocking=True),
            baseline=torch.as_tensor(g.baseline).reshape(1).to(dev, torch.float32, non_blocking=True),
            pos_Tw=up(_bundle(pts, "pos_Tw"), torch.float32), pixel2_uv=up(_bundle(obs, "pixel2_uv"), torch.float32),
            cov_Tw=up(_bundle(pts, "cov_Tw"), torch.float64), pixel2_d=up(_bundle(obs, "pixel2_d"), torch.float32).reshape(-1),
            pixel2_disp=up(_bundle(obs, "pixel2_disp"), torch.float32).reshape(-1),
            pixel2_disp_cov=up(_bundle(obs, "pixel2_disp_cov"), torch.float32).reshape(-1),
            pixel2_uv_cov=up(_bundle(obs, "pixel2_uv_cov"), torch.float32), obs2_covTc=up(_bundle(obs, "obs2_covTc"), torch.float64))
        pose, info = ops.pgo_solve(batch, context["graph_type"], context["lm"])
        host = torch.empty((1, 7), dtype=torch.float64, pin_memory=True)
        host.copy_(pose, non_blocking=True)
        return host, info, batch

    @staticmethod
    def _optimize(context: dict, graph_data: GraphInput):
        host, _, _keep = HIP_TwoFrame_PGO._launch(context, graph_data)
        torch.cuda.current_stream().synchronize()
        return context, GraphOutput(motion=host.clone(), frame_idx=graph_data.frame_idx, from_idx=graph_data.from_idx)

    def start_optimize(self, graph_data: GraphInput) -> None:
        self.has_opt_job = True
        if not self.is_parallel_mode:
            self.context, self.optimize_res = self._optimize(self.context, graph_data)
            return
        side = self.context["stream"]
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            host, info, keep = self._launch(self.context, graph_data)
            done = torch.cuda.Event()
            done.record(side)
        self._pending = (host, done, graph_data, keep, info)

    @property
    def is_running(self) -> bool:
        return self._pending is not None and not self._pending[1].query()

    def get_result(self):
        if self._pending is not None:
            host, done, g, _keep, _ = self._pending
            done.synchronize()
            self.optimize_res = GraphOutput(motion=host.clone(), frame_idx=g.frame_idx, from_idx=g.from_idx)
            self._pending = None
            self.has_opt_job = False
        return self.optimize_res

    get_optimal = get_result

    def write_map(self, global_map) -> None:
        self.write_graph_data(self.get_result(), global_map)

    def write_graph_data(self, result, global_map) -> None:
        if result is None:
            return
        global_map.frames.data["pose"][result.frame_idx] = result.motion[0].double().cpu().float()  # Optimizer.py:104-108

    def terminate(self) -> None:
        self._pending = None


# ----------------------------------------------------------------------------------------------- IFrontend
class HIP_FlowFormerCovFrontend(IFrontend):
    """``FlowFormerCovFrontend`` (Module/Frontend/Frontend.py:143-262) with the hot path in HIP: same YAML args, same call
    structure (``estimate_pair`` concatenates ``A = [L_t2, L_t1]``, ``B = [R_t2, L_t2]`` :219-220, one ``model.inference``,
    ``.float()``), but the window lookups of the network run through ``mv_corr_lookup`` (``install_flowformer_hooks``) and
    ``inference_2_depth`` + ``inference_2_match`` (:183-200: abs, disparity_to_depth(_cov), from_partial_cov) are ONE
    ``mv_frontend_epilogue`` launch instead of ~8 elementwise kernels.

    The network itself stays PyTorch: it is built exactly as the reference does (:147-158) when the FlowFormer package is
    importable; a host that constructs the model itself passes it as ``config.model`` (anything with
    ``inference(imageA, imageB) -> (flow [B,2,H,W], cov = exp(2*log_sigma) [B,2,H,W])``, flownet.py:35-44)."""

    def __init__(self, config: SimpleNamespace):
        super().__init__(config)
        model = getattr(config, "model", None)
        if model is None:
            try:
                from Module.Network.FlowFormer.configs.submission import get_cfg
                from Module.Network.FlowFormerCov import build_flowformer
                from Utility.Utils import reflect_torch_dtype
            except Exception as e:  # noqa: BLE001
                raise ImportError(
                    "HIP_FlowFormerCovFrontend: the FlowFormer network (Module/Network/FlowFormer, the MAC-VO/S_FlowFormer "
                    "submodule) is not importable; initialise the submodule or pass a constructed network as config.model"
                ) from e
            cfg = get_cfg()
            cfg.latentcostformer.decoder_depth = self.config.decoder_depth
            model = build_flowformer(cfg, reflect_torch_dtype(config.enc_dtype), reflect_torch_dtype(config.dec_dtype))
            ckpt = torch.load(self.config.weight, map_location=self.config.device, weights_only=True)
            model.eval()
            model.to(self.config.device)
            model.load_ddp_state_dict(ckpt)
        if hasattr(model, "memory_decoder") or hasattr(model, "memory_encoder"):
            install_flowformer_hooks(model)
        self.model = model

    @property
    def provide_cov(self) -> tuple[bool, bool]:
        return True, True

    def _infer(self, input_A: torch.Tensor, input_B: torch.Tensor):
        input_A = input_A.to(device=self.config.device)
        input_B = input_B.to(device=self.config.device)
        est_flow, est_cov = self.model.inference(input_A, input_B)
        return est_flow.float().contiguous(), est_cov.float().contiguous()

    def _depth_record(self, maps) -> "IStereoDepth.Output":
        return IStereoDepth.Output(depth=maps.depth, cov=maps.depth_cov, disparity=maps.disparity,
                                   disparity_uncertainty=maps.disparity_cov, mask=maps.bad_mask)

    @torch.inference_mode()
    def estimate_depth(self, frame) -> "IStereoDepth.Output":
        flow, cov = self._infer(frame.imageL, frame.imageR)
        maps = ops.frontend_epilogue(flow[0:1], cov[0:1], frame.frame_baseline, frame.fx, cov_is_log=False,
                                     enforce_positive_disparity=self.config.enforce_positive_disparity, want_match=False)
        return self._depth_record(maps)

    @torch.inference_mode()
    def estimate_pair(self, frame_t1, frame_t2):
        flow, cov = self._infer(torch.cat([frame_t2.imageL, frame_t1.imageL], dim=0),
                                torch.cat([frame_t2.imageR, frame_t2.imageL], dim=0))
        maps = ops.frontend_epilogue(flow[0:2], cov[0:2], frame_t2.frame_baseline, frame_t2.fx, cov_is_log=False,
                                     enforce_positive_disparity=self.config.enforce_positive_disparity)
        return self._depth_record(maps), IMatcher.Output(flow=maps.flow, cov=maps.flow_cov, mask=None)

    @torch.inference_mode()
    def estimate_triplet(self, frame_t1, frame_t2):
        flow, cov = self._infer(torch.cat([frame_t1.imageL, frame_t2.imageL, frame_t1.imageL], dim=0),
                                torch.cat([frame_t1.imageL, frame_t2.imageR, frame_t2.imageL], dim=0))
        # NB the reference feeds (L_t1, L_t1) as the first pair (:237-238) — kept as is
        pos = self.config.enforce_positive_disparity
        m1 = ops.frontend_epilogue(flow[0:1], cov[0:1], frame_t1.frame_baseline, frame_t1.fx, cov_is_log=False,
                                   enforce_positive_disparity=pos, want_match=False)
        m2 = ops.frontend_epilogue(flow[1:3], cov[1:3], frame_t2.frame_baseline, frame_t2.fx, cov_is_log=False,
                                   enforce_positive_disparity=pos)
        return self._depth_record(m1), self._depth_record(m2), IMatcher.Output(flow=m2.flow, cov=m2.flow_cov, mask=None)

    @classmethod
    def is_valid_config(cls, config: SimpleNamespace | None) -> None:
        cls._enforce_config_spec(config, {
            "weight": lambda s: isinstance(s, str),
            "device": lambda s: isinstance(s, str) and ("cuda" in s),      # the HIP hot path has no CPU fallback
            "dec_dtype": lambda b: isinstance(b, str) and b in ("fp32", "fp16", "bf16"),
            "enc_dtype": lambda b: isinstance(b, str) and b in ("fp32", "fp16", "bf16"),
            "enforce_positive_disparity": lambda b: isinstance(b, bool),
            "decoder_depth": lambda v: isinstance(v, int),
        })


def _build_flowformer_cov(config: SimpleNamespace, who: str):
    """The network exactly as the reference builds it (StereoDepth.py:143-157, Matching.py:161-176) — or ``config.model``."""
    model = getattr(config, "model", None)
    if model is None:
        try:
            from Module.Network.FlowFormer.configs.submission import get_cfg
            from Module.Network.FlowFormerCov import build_flowformer
            from Utility.Utils import reflect_torch_dtype
        except Exception as e:  # noqa: BLE001
            raise ImportError(f"{who}: the FlowFormer network (Module/Network/FlowFormer, the MAC-VO/S_FlowFormer submodule) is "
                              "not importable; initialise the submodule or pass a constructed network as config.model") from e
        model = build_flowformer(get_cfg(), reflect_torch_dtype(config.enc_dtype), reflect_torch_dtype(config.dec_dtype))
        model.load_ddp_state_dict(torch.load(config.weight, weights_only=True))
        model.to(config.device)
        model.eval()
    if hasattr(model, "memory_decoder") or hasattr(model, "memory_encoder"):
        install_flowformer_hooks(model)
    return model


_FF_COV_SPEC = {
    "weight": lambda s: isinstance(s, str),
    "device": lambda s: isinstance(s, str) and ("cuda" in s),              # the HIP hot path has no CPU fallback
    "enc_dtype": lambda s: s in {"fp16", "bf16", "fp32"},
    "dec_dtype": lambda s: s in {"fp16", "bf16", "fp32"},
}


class HIP_FlowFormerCovDepth(IStereoDepth):
    """``FlowFormerCovDepth`` (Module/Frontend/StereoDepth.py:138-184), the ``IStereoDepth`` of ``FrontendCompose`` configs:
    network in PyTorch (its window lookups through ``mv_corr_lookup``), ``abs`` + ``disparity_to_depth(_cov)`` (:168-171) in
    one ``mv_frontend_epilogue`` launch."""

    def __init__(self, config: SimpleNamespace):
        super().__init__(config)
        self.model = _build_flowformer_cov(config, "HIP_FlowFormerCovDepth")

    @property
    def provide_cov(self) -> bool:
        return True

    @torch.inference_mode()
    def estimate(self, frame) -> "IStereoDepth.Output":
        est_flow, est_cov = self.model.inference(frame.imageL.to(self.config.device), frame.imageR.to(self.config.device))
        m = ops.frontend_epilogue(est_flow.float().contiguous()[0:1], est_cov.float().contiguous()[0:1], frame.frame_baseline,
                                  frame.fx, cov_is_log=False, want_match=False)
        return IStereoDepth.Output(depth=m.depth, cov=m.depth_cov, disparity=m.disparity, disparity_uncertainty=m.disparity_cov)

    @classmethod
    def is_valid_config(cls, config: SimpleNamespace | None) -> None:
        cls._enforce_config_spec(config, _FF_COV_SPEC)


class HIP_FlowFormerCovMatcher(IMatcher):
    """``FlowFormerCovMatcher`` (Module/Frontend/Matching.py:157-197): network in PyTorch with the HIP window lookups;
    ``from_partial_cov`` (:34-40) pads the two variance channels with a zero sigma_uv plane."""

    def __init__(self, config: SimpleNamespace):
        super().__init__(config)
        self.model = _build_flowformer_cov(config, "HIP_FlowFormerCovMatcher")

    @property
    def provide_cov(self) -> bool:
        return True

    def forward(self, frame_t1, frame_t2) -> "IMatcher.Output":
        flow, flow_cov = self.model.inference(frame_t1.imageL.to(self.config.device), frame_t2.imageL.to(self.config.device))
        return IMatcher.Output.from_partial_cov(flow=flow, cov=flow_cov)

    @classmethod
    def is_valid_config(cls, config: SimpleNamespace | None) -> None:
        cls._enforce_config_spec(config, _FF_COV_SPEC)


class HIP_CUDAGraph_FlowFormerCovFrontend(HIP_FlowFormerCovFrontend):
    """``CUDAGraph_FlowFormerCovFrontend`` (Module/Frontend/Frontend.py:264-353): the joint stereo + temporal inference of
    ``estimate_pair`` is captured once (3 warm-up runs on a side stream, then ``torch.cuda.graph`` = a hipGraph on ROCm)
    and replayed per frame on static input buffers; outputs are cloned (:344-347).  The HIP window lookups installed by
    ``install_flowformer_hooks`` are plain stream launches through the C ABI, so they are captured with the network's own
    kernels; the depth / match epilogue stays one ``mv_frontend_epilogue`` launch after the replay.  ``estimate_depth`` and
    ``estimate_triplet`` are inherited (eager), as in the reference."""

    def __init__(self, config: SimpleNamespace):
        super().__init__(config)
        assert "cuda" in self.config.device.lower(), "HIP_CUDAGraph_FlowFormerCovFrontend can only run on a GPU device."
        self.cuda_graph = None
        torch.backends.cuda.matmul.allow_tf32 = True          # the reference's settings (:275-277)
        torch.backends.cudnn.allow_tf32 = True
        torch.set_float32_matmul_precision("medium")

    def cuda_graph_estimate(self, inp_A: torch.Tensor, inp_B: torch.Tensor):
        if self.cuda_graph is None:
            static_A, static_B = torch.empty_like(inp_A, device="cuda"), torch.empty_like(inp_B, device="cuda")
            static_A.copy_(inp_A)
            static_B.copy_(inp_B)
            out_val = out_cov = None
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    out_val, out_cov = self.model.inference(static_A, static_B)
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_flow, static_cov = self.model.inference(static_A, static_B)
            self.cuda_graph = SimpleNamespace(graph=graph, shape=inp_A.shape, input_A=static_A, input_B=static_B,
                                              flow=static_flow, flow_cov=static_cov)
            return out_val, out_cov
        g = self.cuda_graph
        assert inp_A.shape == g.shape, f"Input shape mismatch for graph replay: {inp_A.shape} != {g.shape}"
        g.input_A.copy_(inp_A)
        g.input_B.copy_(inp_B)
        g.graph.replay()
        return g.flow.clone(), g.flow_cov.clone()

    @torch.inference_mode()
    def estimate_pair(self, frame_t1, frame_t2):
        input_A = torch.cat([frame_t2.imageL, frame_t1.imageL], dim=0).to(device=self.config.device)
        input_B = torch.cat([frame_t2.imageR, frame_t2.imageL], dim=0).to(device=self.config.device)
        est_flow, est_cov = self.cuda_graph_estimate(input_A, input_B)
        flow, cov = est_flow.float().contiguous(), est_cov.float().contiguous()
        maps = ops.frontend_epilogue(flow[0:2], cov[0:2], frame_t2.frame_baseline, frame_t2.fx, cov_is_log=False,
                                     enforce_positive_disparity=self.config.enforce_positive_disparity)
        return self._depth_record(maps), IMatcher.Output(flow=maps.flow, cov=maps.flow_cov, mask=None)


# ----------------------------------------------------------------------------------------------- FlowFormer hooks
class _FusedProjForward:
    """``forward`` of a FlowFormer ``PatchEmbed.proj`` (``nn.Sequential`` of three 6x6 stride-2 ``Conv2d``) through ``mv_cost_patch_embed_t``.

    * The Sequential itself is left in place (``proj.forward`` is rebound, nothing is re-parented): ``state_dict`` keys, ``.to()``, ``load_state_dict``
      keep working, and the packed 16-bit weight fragments are rebuilt whenever a parameter's storage or version counter changes.
    * Default numerics policy: the fused kernel rounds slices, weights and both intermediate maps to a 16-bit type, so it is used when the encoder
      already runs in one — fp16 / bf16 slices (autocast, ``MACVO_Fast.yaml:73-74``) with operands of that same type, cells read and tokens written in
      it, no widening on either side.  fp32 slices (``Paper_Reproduce``-style configurations: PyTorch-ROCm convolutions are full fp32) keep the
      original layers unless ``force`` (``install_flowformer_hooks(fuse_patch_embed=True)`` / ``MACVO_HIP_PATCH_EMBED=1``) opts in to IEEE-half
      operands (11 significant bits, ~2.5e-4 of the output scale; conversions saturate at +-65504 instead of overflowing).
    * Training / autograd: falls through to the layers whenever a gradient could be required."""

    def __init__(self, proj: torch.nn.Sequential, force: bool = False):
        self.proj, self.force = proj, force
        self._packed: dict = {}        # operand -> (key, PatchEmbedWeights)

    def _key(self):
        def version(p):
            try:
                return p._version
            except RuntimeError:        # inference tensors (a model built under torch.inference_mode) carry no version counter; they cannot be
                return -1               # updated in place either, so storage + dtype identify them
        return tuple((p.data_ptr(), version(p), p.dtype) for p in self.proj.parameters())

    def repack(self, operand: str = "f16"):
        key = self._key()
        hit = self._packed.get(operand)
        if hit is None or hit[0] != key:
            hit = (key, ops.PatchEmbedWeights.from_proj(self.proj, operand=operand))
            self._packed[operand] = hit
        return hit[1]

    def __call__(self, x):                                      # x: [S, 1, H2p, W2p] — PatchEmbed.forward has already padded it
        layers = torch.nn.Sequential.forward
        if not (x.is_cuda and x.dim() == 4 and x.shape[1] == 1 and ops.cost_patch_embed_supported(x.shape[-2], x.shape[-1])):
            return layers(self.proj, x)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.proj.parameters())):
            return layers(self.proj, x)
        if x.dtype == torch.float16:
            operand = "f16"
        elif x.dtype == torch.bfloat16:
            operand = "bf16"
        elif x.dtype == torch.float32 and self.force:
            operand = "f16"
        else:
            return layers(self.proj, x)
        if x.data_ptr() % 16 or not x.is_contiguous():          # (a sliced / offset view: the kernel wants 16-byte aligned slices)
            return layers(self.proj, x)
        try:
            out = ops.cost_patch_embed(x, self.repack(operand))  # result in x.dtype, as Conv2d returns it outside autocast
        except ops.L.MacvoHipError:                              # anything the kernel refuses falls through to the layers, as documented
            return layers(self.proj, x)
        if torch.is_autocast_enabled():                          # ... and in the autocast dtype under autocast, as the Conv2d stack would
            out = out.to(torch.get_autocast_dtype("cuda"))
        return out


def install_flowformer_hooks(model, volume_precision: str | None = None, fuse_patch_embed: bool | None = None) -> list[str]:
    """Route the three frontend kernels of ``FlowFormerCov`` through the HIP library by rebinding bound methods on the model
    instance — signatures and result layouts are those of the methods they replace, the network's own code is untouched:

    * ``memory_decoder.encode_flow_token(cost_maps, coords)`` — the 9x9 window lookup, once per decoder iteration
      (Module/Network/FlowFormerCov/covhead.py:92) -> ``ops.corr_lookup`` (``[B, 81, H1, W1]`` fp32);
    * ``memory_decoder.upsample_flow(flow, mask)`` — the convex 8x upsampling of flow and log-sigma, twice per iteration
      (covhead.py:124-126, 133-135; both callers hand over fp32 and an already scaled mask) -> ``ops.convex_upsample``;
    * ``memory_encoder.corr(fmap1, fmap2)`` — the all-pairs volume (``einsum('bhid,bhjd->bhij')`` of FlowFormer's
      ``MemoryEncoder``, reached from flownet.py:26) -> ``ops.corr_volume``; result ``[B, heads = 1, H1, W1, H2, W2]`` in the
      dtype of the feature maps, as the einsum would return it (fp32 features: the kernel's fp32 output as is; 16-bit
      encoder dtypes: one cast, exactly the rounding the einsum's 16-bit output has — flownet.py:27 widens it again).

    ``volume_precision``: None = ``ops.default_volume_precision()`` ("f16x2" unless ``MACVO_HIP_VOLUME_PRECISION`` says otherwise) — the
    same default as ``pipeline.HotPathConfig`` and ``bench.py``; the plugins' YAML key sets are the reference's, so they take the default.

    * ``memory_encoder...patch_embed.proj`` — the cost patch embedding's three convolutions (row (f)2) -> ``ops.cost_patch_embed``
      (``_FusedProjForward``).  ``fuse_patch_embed``: None = for 16-bit slices only (the encoder's own precision is kept; fp32 slices stay on the
      PyTorch layers), True = also for fp32 slices (IEEE-half operands), False = never; ``MACVO_HIP_PATCH_EMBED=1|0`` sets the default.

    Every attribute that exists is rebound (the FlowFormer submodule is absent from some checkouts, and a stand-in model may
    carry only part of them); the names of the rebound methods are returned so that a caller can insist on all three."""
    done = []
    if volume_precision is None:
        volume_precision = ops.default_volume_precision()
    dec = getattr(model, "memory_decoder", None)
    if dec is not None and hasattr(dec, "encode_flow_token"):
        dec.encode_flow_token = lambda cost_maps, coords: ops.corr_lookup(
            cost_maps if cost_maps.dtype == torch.float16 else cost_maps.float(), coords.float(), 4)   # an fp16 volume is read as it is
        done.append("memory_decoder.encode_flow_token")
    if dec is not None and hasattr(dec, "upsample_flow"):
        dec.upsample_flow = lambda flow, mask: ops.convex_upsample(      # a 16-bit mask (autocast) is read as it is: no widened 22-MB copy
            flow.float(), mask if mask.dtype in (torch.float16, torch.bfloat16) else mask.float(), 1.0, False)
        done.append("memory_decoder.upsample_flow")
    enc = getattr(model, "memory_encoder", None)
    if enc is not None and hasattr(enc, "corr"):
        heads = int(getattr(getattr(enc, "cfg", None), "cost_heads_num", 1) or 1)
        if heads != 1:
            raise ops.L.MacvoHipError("install_flowformer_hooks: cost_heads_num != 1 (every MAC-VO config uses 1, Config/Train/Demo.yaml)")

        def corr(fmap1, fmap2):
            B, _, H, W = fmap1.shape
            if fmap1.dtype in (torch.float16, torch.bfloat16):
                # Fast mode: the einsum's 16-bit result with its one rounding in the GEMM epilogue (two 10-MB NHWC copies of the feature
                # maps in front of it; round 3: fp32 kernel output -> cast -> the model's .float(): three passes over 184 MB)
                vol = ops.corr_volume_out16(fmap1.permute(0, 2, 3, 1).contiguous(), fmap2.permute(0, 2, 3, 1).contiguous())
                if vol is not None:
                    return vol.view(B, 1, H, W, fmap2.shape[2], fmap2.shape[3])
            vol = ops.corr_volume(fmap1.contiguous(), fmap2.contiguous(), layout="chw",
                                  precision=volume_precision if fmap1.dtype == torch.float32 else "exact")
            vol = vol if vol.dtype == fmap1.dtype else vol.to(fmap1.dtype)
            return vol.view(B, 1, H, W, fmap2.shape[2], fmap2.shape[3])

        enc.corr = corr
        done.append("memory_encoder.corr")
    # (f)2: the cost patch embedding.  FlowFormer's cost encoder owns `patch_embed = PatchEmbed(patch_size 8, in_chans 1, embed_dim 64)` whose
    # `proj` (three 6x6 stride-2 convolutions) eats the whole volume slice by slice: `proj.forward` is rebound to the fused kernel for the slice
    # sizes it covers; any other size, dtype or mode falls through to the original layers.
    if fuse_patch_embed is None:
        env = os.environ.get("MACVO_HIP_PATCH_EMBED", "auto").lower()
        fuse_patch_embed = {"1": True, "on": True, "true": True, "0": False, "off": False, "false": False}.get(env)
    pe, pe_name = None, ""
    if fuse_patch_embed is not False and enc is not None and hasattr(enc, "named_modules"):
        # public FlowFormer: memory_encoder.cost_perceiver_encoder.patch_embed; accept it directly under the encoder too
        for name, mod in enc.named_modules():
            if (name == "patch_embed" or name.endswith(".patch_embed")) and isinstance(getattr(mod, "proj", None), torch.nn.Sequential):
                pe, pe_name = mod, name
                break
    if pe is not None:
        fused = _FusedProjForward(pe.proj, force=fuse_patch_embed is True)
        try:
            fused.repack()                                  # validates the layer shapes now; a stack that is not patch_size 8 keeps its layers
        except ops.L.MacvoHipError:
            fused = None
        if fused is not None:
            # an instance attribute shadows nn.Sequential.forward: the module, its parameters and its state_dict keys stay where they were
            pe.proj.forward = fused
            done.append(f"memory_encoder.{pe_name}.proj")
    if not done:
        raise ops.L.MacvoHipError("install_flowformer_hooks: the model has none of memory_decoder.encode_flow_token / "
                                  "upsample_flow / memory_encoder.corr")
    return done


def patch_reference_correlation(correlation=None) -> list[str]:
    """Make the reference's in-tree PWC-Net family call the HIP local correlation: rebind ``FunctionCorrelation`` in every
    reference module that imported it by name (``Module/Network/PWCNet/RAFTCov.py:7``, ``pwc/pwc_model.py:12``,
    ``pwc/pwc_model_tartanvo.py:15``) and in ``pwc/correlation.py`` itself.  The reference's module imports ``cupy`` at the
    top (correlation.py:5) only to JIT its CUDA kernels; on a ROCm box without cupy an inert placeholder is registered first
    (the kernels are never launched once the function is rebound).  Returns the names of the patched modules."""
    import importlib
    import sys
    import types

    if "cupy" not in sys.modules:
        try:
            importlib.import_module("cupy")
        except Exception:  # noqa: BLE001 - absent or CUDA-only build
            stub = types.ModuleType("cupy")
            stub.cuda = types.SimpleNamespace(compile_with_cache=lambda *a, **k: None)
            stub.memoize = lambda **k: (lambda f: f)
            sys.modules["cupy"] = stub
    fn = correlation or FunctionCorrelation
    patched = []
    for name in ("Module.Network.PWCNet.pwc.correlation", "Module.Network.PWCNet.pwc.pwc_model",
                 "Module.Network.PWCNet.pwc.pwc_model_tartanvo", "Module.Network.PWCNet.RAFTCov"):
        try:
            m = importlib.import_module(name)
        except ImportError:
            continue
        m.FunctionCorrelation = fn
        patched.append(name)
    return patched


class HIP_TartanVOCovMatcher(IMatcher):
    """``TartanVOCovMatcher`` (Module/Frontend/Matching.py:233-274): the reference's in-tree ``RAFTFlowCovNet``
    (Module/Network/PWCNet/RAFTCov.py:45-107 — PWC-Net feature pyramid + GaussianGRU covariance head) stays PyTorch-ROCm,
    its five 81-channel local correlations per frame (pwc_model.py:178-233; the reference's only hand-written CUDA kernel,
    JIT-compiled through cupy and unavailable on ROCm) run in ``mv_local_corr81``.  Same YAML ``args`` (weight, device);
    ``weight: ""`` keeps the random initialisation (smoke tests / benchmarks without the release checkpoint).
    The network source lives in the MAC-VO checkout: outside one the constructor raises."""

    correlation = None      # injectable (tests run the wiring on the CPU with the oracle's definition)

    def __init__(self, config: SimpleNamespace):
        super().__init__(config)
        patched = patch_reference_correlation(type(self).correlation)
        if "Module.Network.PWCNet.RAFTCov" not in patched:
            raise ops.L.MacvoHipError("HIP_TartanVOCovMatcher needs the MAC-VO checkout on sys.path (Module.Network.PWCNet)")
        from Module.Network.PWCNet import RAFTFlowCovNet  # type: ignore

        cfg = SimpleNamespace(decoder="raft", dim=64, dropout=0.1, num_heads=4, mixtures=4, gru_iters=12, kernel_size=3)
        model = RAFTFlowCovNet(cfg, self.config.device)                          # Matching.py:242-248
        if self.config.weight:
            model.load_ddp_state_dict(torch.load(self.config.weight, map_location="cpu", weights_only=True))
        self.model = model.to(self.config.device).eval()

    @property
    def provide_cov(self) -> bool:
        return True

    def forward(self, frame_t1, frame_t2) -> "IMatcher.Output":
        flow, flow_cov = self.model.inference(frame_t1.imageL, frame_t2.imageL)
        # Matching.py:260-268 as written: the validity mask marks the un-padded centre; with no padding (the network resizes
        # back to the input size) the slice 0:-0 is empty and the mask stays all-False — reference behaviour, kept
        mask = torch.zeros_like(flow[:, :1], dtype=torch.bool)
        pad_h = (frame_t1.height - flow.size(-2)) // 2
        pad_w = (frame_t1.width - flow.size(-1)) // 2
        mask[..., pad_h:-pad_h, pad_w:-pad_w] = True
        flow = _pad_to(flow, frame_t1.height, frame_t1.width)
        flow_cov = _pad_to(flow_cov, frame_t1.height, frame_t1.width)
        return IMatcher.Output.from_partial_cov(flow=flow, cov=flow_cov, mask=mask)

    @classmethod
    def is_valid_config(cls, config: SimpleNamespace | None) -> None:
        cls._enforce_config_spec(config, {"weight": lambda s: isinstance(s, str), "device": _is_device})


def _pad_to(x: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """``padTo(x, (H, W), dim=(-2, -1), value=nan)`` (Utility/Utils.py:95-125): symmetric NaN padding, even amounts only."""
    ph, pw = H - x.size(-2), W - x.size(-1)
    assert ph % 2 == 0 and pw % 2 == 0, "Can only handle even padding."
    if ph == 0 and pw == 0:
        return x
    return torch.nn.functional.pad(x, (pw // 2, pw // 2, ph // 2, ph // 2), mode="constant", value=float("nan"))


def FunctionCorrelation(tenFirst, tenSecond):
    """Drop-in for ``Module/Network/PWCNet/pwc/correlation.py:372-373`` (forward / inference only — the gradient kernels
    :105-233 are training code): same argument names, same contiguity asserts (:283-284), HIP kernel underneath.  Patch
    with ``Module.Network.PWCNet.pwc.correlation.FunctionCorrelation = macvo_amd.plugins.FunctionCorrelation``."""
    import torch

    from . import ops

    assert tenFirst.is_contiguous()
    assert tenSecond.is_contiguous()
    if not tenFirst.is_cuda:
        raise NotImplementedError()          # as the reference (:323-324): there is no CPU path
    if torch.is_grad_enabled() and (tenFirst.requires_grad or tenSecond.requires_grad):
        raise NotImplementedError("macvo_amd FunctionCorrelation is forward-only; wrap the call in torch.no_grad()")
    return ops.local_corr81(tenFirst.float(), tenSecond.float())
