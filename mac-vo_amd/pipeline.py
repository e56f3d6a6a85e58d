"""The per-frame hot path in the reference's call order (``Odometry/MACVO.py:173-311``), on one GPU.

``HotPath.step`` is what ``MACVO.run_pair`` executes between "the learned layers produced feature maps /
GRU updates" and "the optimised pose is written back", with every arithmetic step in the HIP kernels:

    Frontend.estimate_pair   MACVO.py:182   corr volume (A5) -> 12 x window lookup (A6) -> epilogue (A8/A2)
    KeypointSelector         MACVO.py:197   dense selector (A10/A11) + host randperm (bit-exact indices)
    gathers / tracking       MACVO.py:198-232  kp_track (A12)
    pixel2point_NED, world   MACVO.py:240,273-281  backproject (A12/A16)
    ObsCovModel.estimate x2  MACVO.py:241-242  match_cov (A13-A16), world rotation fused
    OutlierFilter.filter     MACVO.py:269   obs_filter (validity mask instead of row compaction)
    Optimizer                MACVO.py:309-311  pgo_solve (A17-A22), pose written back in fp32 (A22)

The learned FlowFormer layers (Twins encoder, cost-token transformer, GRU) are NOT part of this module: their
outputs for a frame arrive as :class:`FrameInputs` already resident in HBM (the network "stays
PyTorch-ROCm"; its source and weights are absent from the reference checkout).

State carried between frames lives on the device: the previous frame's depth maps and the previous pose
(StaticMotionModel: the prior of frame t is the optimised pose of frame t-1, MotionModel.py:134-142).
The only host synchronisation per frame is the selector's candidate count (needed by ``torch.randperm``).
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass, field

import torch

from . import ops

_RANGES = os.environ.get("MV_TRACE_RANGES", "0") not in ("", "0")


def traced(name: str):
    """Opt-in (``MV_TRACE_RANGES=1``) profiler range around a pipeline stage, named like the reference's ``Timer`` scopes
    (``Frontend.estimate``: ``Module/Frontend/Frontend.py:215-216``; ``Odom_Runtime``: ``Odometry/MACVO.py:350-351``).  On ROCm
    ``torch.cuda.nvtx`` emits roctx ranges (``rocprofv3 --marker-trace``).  Off by default: two host calls per stage per frame."""
    def deco(fn):
        if not _RANGES:
            return fn
        import functools

        @functools.wraps(fn)
        def wrapped(*a, **k):
            torch.cuda.nvtx.range_push(name)
            try:
                return fn(*a, **k)
            finally:
                torch.cuda.nvtx.range_pop()
        return wrapped
    return deco


@dataclass
class Camera:
    fx: float
    fy: float
    cx: float
    cy: float
    baseline: float
    H: int
    W: int

    @property
    def K4(self):
        return (self.fx, self.fy, self.cx, self.cy)


@dataclass
class HotPathConfig:
    """Values of ``Config/Experiment/MACVO/MACVO_Fast.yaml`` (:22-104) that touch the hot path."""
    num_point: int = 200
    edgewidth: int = 32
    match_cov_default: float = 0.25
    selector: str = "nodepth"            # CovAwareSelector_NoDepth | "full" = CovAwareSelector
    kp_kernel_size: int = 7
    kp_mask_width: int = 32
    max_match_cov: float = 100.0
    max_depth_cov: float = 250.0
    max_depth: float | str = "auto"      # "auto" -> fx * baseline (KeypointSelector.py:263)
    cov_kernel_size: int = 31
    min_flow_cov: float = 0.25
    min_depth_cov: float = 0.05
    graph_type: str = "disp"
    min_num_point: int = 10              # MACVO.py:64
    filters: int = ops.FILTER_COV_SANITY  # CovarianceSanityFilter
    filter_min_depth: float = 0.05
    radius: int = 4
    feature_layout: str = "chw"
    use_graphs: bool = False             # hipGraph-replay the decoder-side segment for inputs marked `static`
    mapping: bool = False                # dense-mapping tail of run_pair (MACVO.py:313-337; `mapping: true` in MACVO_Fast)
    map_num_point: int = 2000            # :315
    map_max_depth: float = 5.0           # MappingPointSelector args (Config/Experiment/MACVO/MACVO_Fast.yaml)
    map_max_depth_cov: float = 0.005
    map_mask_width: int = 32
    # fp32 features: "f16x2" (the default everywhere, ops.default_volume_precision(): per-row power-of-two scales + two fp16 pieces,
    # three products on the 16-bit matrix pipe, error <= ~2^-21 sum |a||b|) | "bf16x3" three bf16 pieces, six products | "exact" fp32
    # MFMA (bitwise fmaf chain) | "split3" / "split2" the round-1 tile kernels over pre-split planes (layout "hwc").  Shapes the
    # streaming split kernel does not cover run the exact kernel.  16-bit features ignore it.
    volume_precision: str = field(default_factory=ops.default_volume_precision)
    # 16-bit features: "fp32" = fp32 cells (rounds 1-3) | "encoder" = the volume in the features' fp16 type, one rounding in the GEMM's epilogue —
    # what the reference's Fast mode computes (einsum of fp16 maps, flownet.py:26-27; MACVO_Fast.yaml:73-74) at half the bytes; the lookups read
    # the 2-byte cells.  fp16 features in layout "hwc" with C = 128 / 256 (native driver); anything else keeps fp32 cells.
    volume_store: str = "fp32"
    async_backend: bool | None = None    # native driver: issue a frame's backend launches from a second host thread (None: the
                                         # library's default / MV_PIPE_ASYNC_BACKEND); identical results either way


@dataclass
class FrameInputs:
    """What the learned layers hand to the hot path for one ``estimate_pair`` (all GPU-resident).

    fmap1 / fmap2 : ``[2, C, H/8, W/8]`` (layout "chw") or ``[2, H/8, W/8, C]`` ("hwc"); pair 0 = stereo
                    (L_t2 vs R_t2), pair 1 = temporal (L_t1 vs L_t2)  (Frontend.py:219-220)
    coords        : ``[iters, 2, 2, H/8, W/8]`` fp32 — coords1 entering each decoder iteration (covhead.py:85-92)
    flow, logcov  : ``[2, 2, H, W]`` fp32 — last upsampled flow / log-sigma predictions (covhead.py:140)
    """
    fmap1: torch.Tensor
    fmap2: torch.Tensor
    coords: torch.Tensor
    flow: torch.Tensor | None = None
    logcov: torch.Tensor | None = None
    # alternative to flow/logcov: the last decoder iteration's 1/8-resolution fields and convex
    # upsampling masks (covhead.py:119-135); the hot path then runs mv_convex_upsample (+ fused exp(2*cov)) itself
    flow8: torch.Tensor | None = None        # [2, 2, H/8, W/8]  coords1 - coords0
    cov8: torch.Tensor | None = None         # [2, 2, H/8, W/8]  cov_coords1 - cov_coords0
    up_mask: torch.Tensor | None = None      # [2, 576, H/8, W/8] flow branch mask BEFORE the 0.25 scale (:121)
    cov_mask: torch.Tensor | None = None     # [2, 576, H/8, W/8] log-sigma branch mask (0.25 already applied, :41)
    # event recorded by whoever produced fmap1/fmap2 (None = already complete, e.g. resident inputs): the volume GEMM
    # runs on its own stream and must not start before its operands exist
    ready: "torch.cuda.Event | None" = None
    # previous LEFT image [1|-,3,H,W] in [0,1] for the map-point colours (MACVO.py:326-328); optional, mapping mode only
    image: torch.Tensor | None = None
    # frame timestamp (StereoData.frame_ns) — recorded in the device-resident map when one is attached
    time_ns: int = 0
    # promise that every tensor above lives at a fixed address for the lifetime of the HotPath (e.g. the static output
    # buffers of a graph-captured network, as in the reference's CUDAGraph frontend): allows hipGraph replay
    static: bool = False


@dataclass
class FrameResult:
    pose: torch.Tensor                 # [7] fp32 GPU — optimised pose of this frame (write_graph_data)
    pose_f64: torch.Tensor | None      # [1,7] fp64 GPU
    info: torch.Tensor | None          # [1,4] fp64 GPU {loss, steps, rejects, loss0}
    kp0_uv: torch.Tensor | None        # [n,2] int64 GPU selected keypoints
    n_valid: torch.Tensor | None       # [1] int32 GPU surviving observations
    extras: dict = field(default_factory=dict)
    map_points: "ops.MapPoints | None" = None   # mapping mode: the frame's dense map points (valid after sync_pose())


class HotPath:
    def __init__(self, cam: Camera, cfg: HotPathConfig | None = None, device: str | torch.device = "cuda",
                 keep_extras: bool = False):
        self.cam, self.cfg = cam, cfg or HotPathConfig()
        self.dev = torch.device(device)
        self.keep_extras = keep_extras
        self.lm = ops.lm_default_params()
        self.maps_prev_for_next: ops.FrontendMaps | None = None   # depth maps of the newest frontend'ed frame
        self._side = torch.cuda.Stream(device=self.dev, priority=-1)      # PGO stream
        self._back = torch.cuda.Stream(device=self.dev, priority=-1)      # pose-dependent half of a frame (tracking .. filter)
        self._perm_pinned = [torch.empty((max(self.cfg.num_point, 1),), dtype=torch.int64, pin_memory=True) for _ in range(4)]
        self._perm_slot = 0
        self._graphs: dict = {}                   # (id(inputs), slot) -> captured decoder-side segment
        self._frame_no = 0
        self._pgo_done = None
        self._pgo_keep = None
        self._map_done = None
        self._prev_image = None
        self.pose = torch.tensor([0, 0, 0, 0, 0, 0, 1], dtype=torch.float32, device=self.dev)
        c = self.cfg
        self._max_depth = cam.fx * cam.baseline if c.max_depth == "auto" else float(c.max_depth)
        self._intr = torch.tensor([cam.K4], dtype=torch.float32, device=self.dev)
        self._bl = torch.tensor([cam.baseline], dtype=torch.float32, device=self.dev)
        self._vols = [None, None]                 # double-buffered cost volumes (184 MB each @640x480, B = 2)
        self._vol_free = [None, None]             # event: the last reader (lookups) of that buffer has finished
        self._vol_idx = 0
        # the GEMM stream gets the LOWEST priority: whenever CU slots free up, the small latency-bound kernels of the other
        # streams (lookups, selector, backend, PGO) should be dispatched first and the GEMM fills whatever is left
        self._vol_stream = torch.cuda.Stream(device=self.dev, priority=0)
        self._tok = None
        self.last_tokens = None
        # offsets table: row n = [0, n] (one problem of n points) — avoids an H2D copy per frame
        m = self.cfg.num_point + 1
        self._offs = torch.stack([torch.zeros(m, dtype=torch.int32), torch.arange(m, dtype=torch.int32)], 1).to(self.dev)

    # ------------------------------------------------------------------ frontend part of the hot path
    def _volume(self, x: FrameInputs):
        """The MFMA-bound volume GEMM of THIS frame runs on its own stream and overlaps the latency-bound decoder-side
        work (lookups, selector, backend) of the PREVIOUS frame that is still queued on the other streams — the two
        frontends are independent (Frontend.py:219-224).  Two volume buffers alternate; a buffer is rewritten only
        after the lookups that read it have finished."""
        c = self.cfg
        main = torch.cuda.current_stream()
        slot = self._frame_no % 6
        self._frame_no += 1
        k = slot & 1
        n_rows = x.fmap1.shape[0] * x.coords.shape[-1] * x.coords.shape[-2]
        if self._vols[k] is not None and self._vols[k].shape[0] != n_rows:
            self._vols[k] = None
        vs = self._vol_stream
        if x.ready is not None:
            vs.wait_event(x.ready)            # producer of the feature maps (the encoder) signals readiness
        if self._vol_free[k] is not None:
            vs.wait_event(self._vol_free[k])
        with torch.cuda.stream(vs):
            self._vols[k] = ops.corr_volume(x.fmap1, x.fmap2, layout=c.feature_layout, out=self._vols[k],
                                            precision=c.volume_precision if x.fmap1.dtype == torch.float32 else "exact")
            vol_done = torch.cuda.Event()
            vol_done.record(vs)
        main.wait_event(vol_done)
        return self._vols[k], k, slot

    def _decoder_side(self, x: FrameInputs, vol: torch.Tensor, with_selector: bool):
        """12 x window lookup + (convex upsampling) + epilogue + dense selector stage: everything between the volume and
        the host's randperm.  Pure enqueue (graph-capturable)."""
        c, cam = self.cfg, self.cam
        tok = None
        for it in range(x.coords.shape[0]):
            tok = ops.corr_lookup(vol, x.coords[it], c.radius, out=tok)
        if x.flow8 is not None:
            flow = ops.convex_upsample(x.flow8, x.up_mask, mask_scale=0.25)
            cov = ops.convex_upsample(x.cov8, x.cov_mask, mask_scale=1.0, exp2_out=True)     # exp(2*cov) fused
            maps = ops.frontend_epilogue(flow, cov, cam.baseline, cam.fx, cov_is_log=False)
        else:
            maps = ops.frontend_epilogue(x.flow, x.logcov, cam.baseline, cam.fx, cov_is_log=True)
        cands = None
        if with_selector:
            cands = ops.kp_select("nodepth", cam.H, cam.W, flow_cov=maps.flow_cov, kernel_size=c.kp_kernel_size,
                                  mask_width=c.kp_mask_width, max_match_cov=c.max_match_cov)
        return tok, maps, cands

    def frontend(self, x: FrameInputs, with_selector: bool = False):
        """volume -> decoder side; returns (maps, cands | None, host_count | None).  With ``use_graphs`` and a ``static``
        input the decoder side is one hipGraph replay (12 lookups + epilogue + selector + count copy = 17 nodes)."""
        c = self.cfg
        vol, k, slot = self._volume(x)
        main = torch.cuda.current_stream()
        graphable = c.use_graphs and x.static and with_selector and c.selector == "nodepth"
        host_count = None
        if graphable:
            key = (id(x), slot)
            g = self._graphs.get(key)
            if g is None:
                # eager warm-up (lazy initialisations must not happen under capture), then capture
                self._decoder_side(x, vol, True)
                hc = torch.empty((4,), dtype=torch.int32, pin_memory=True)
                main.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    tok, maps, cands = self._decoder_side(x, vol, True)
                    hc.copy_(cands.count, non_blocking=True)
                g = (graph, tok, maps, cands, hc, x, vol)
                self._graphs[key] = g
            graph, tok, maps, cands, host_count = g[:5]
            graph.replay()
            cands._n = None
        else:
            tok, maps, cands = self._decoder_side(x, vol, with_selector and c.selector == "nodepth")
            if cands is not None:
                host_count = torch.empty((4,), dtype=torch.int32, pin_memory=True)
                host_count.copy_(cands.count, non_blocking=True)
        free = torch.cuda.Event()
        free.record(main)
        self._vol_free[k] = free
        self.last_tokens = tok
        return maps, cands, host_count

    def initialize(self, x: FrameInputs, init_pose: torch.Tensor | None = None) -> None:
        """Frame 0: ``MACVO.initialize`` (:158-171) — depth only, pose = prior."""
        self.maps_prev_for_next = self.frontend(x)[0]
        self._prev_image = x.image
        if init_pose is not None:
            self.pose = init_pose.to(self.dev, torch.float32).reshape(7).clone()

    # ------------------------------------------------------------------ one run_pair, in two halves
    @traced("Frontend.estimate")
    def enqueue_frontend(self, x: FrameInputs) -> "_Pending":
        """Everything of a frame that does not depend on the previous pose: volume, lookups, epilogue and the dense
        selector stage.  Only enqueues work; the candidate count travels to a pinned host word behind an event, so a
        later ``finish`` waits for THIS frame's selector and not for whatever was queued after it."""
        c, cam = self.cfg, self.cam
        maps0 = self.maps_prev_for_next
        maps1, cands, host_count = self.frontend(x, with_selector=True)
        if cands is None:   # CovAwareSelector needs the previous frame's depth maps: eager, after the frontend
            cands = ops.kp_select("full", cam.H, cam.W, flow_cov=maps1.flow_cov, depth0=maps0.depth,
                                  depth0_cov=maps0.depth_cov, depth1=maps1.depth, depth1_cov=maps1.depth_cov,
                                  kernel_size=c.kp_kernel_size, mask_width=c.kp_mask_width, max_depth=self._max_depth,
                                  max_depth_cov=c.max_depth_cov, max_match_cov=c.max_match_cov)
            host_count = torch.empty((4,), dtype=torch.int32, pin_memory=True)
            host_count.copy_(cands.count, non_blocking=True)
        cands_m = host_count_m = None
        if c.mapping:   # MappingPointSelector works on the PREVIOUS frame's depth maps (KeypointSelector.py:87-97): count travels with the other one
            cands_m = ops.kp_select("mapping", cam.H, cam.W, depth0=maps0.depth, depth0_cov=maps0.depth_cov,
                                    mask_width=c.map_mask_width, max_depth=c.map_max_depth, max_depth_cov=c.map_max_depth_cov)
            host_count_m = torch.empty((4,), dtype=torch.int32, pin_memory=True)
            host_count_m.copy_(cands_m.count, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.maps_prev_for_next = maps1
        pend = _Pending(maps0, maps1, cands, host_count, ev)
        pend.cands_m, pend.host_count_m, pend.image0 = cands_m, host_count_m, self._prev_image
        self._prev_image = x.image
        return pend

    @traced("Odom_Runtime")
    def finish(self, pend: "_Pending", pose_sink: torch.Tensor | None = None) -> FrameResult:
        """Host randperm (bit-exact indices) + the pose-dependent half: tracking, back-projection, covariances, filter,
        PGO.  The solve runs on a side stream (the GPU analogue of the reference's optimizer child process,
        Optimization/Interface.py:80-96): the next frame's frontend overlaps it, the next frame's back-projection
        waits for it."""
        c, cam = self.cfg, self.cam
        maps0, maps1, cands = pend.maps0, pend.maps1, pend.cands
        pend.event.synchronize()
        cands._n = int(pend.host_count[0])
        back, side = self._back, self._side
        # The backend runs on its own stream: it must not queue behind the NEXT frame's decoder-side work that
        # enqueue_frontend already put on the main stream (that work waits for the next volume GEMM).
        back.wait_event(pend.event)
        with torch.cuda.stream(back):
            self._perm_slot = (self._perm_slot + 1) % len(self._perm_pinned)
            kp0 = cands.finish(c.num_point, staging=self._perm_pinned[self._perm_slot])  # CPU randperm, as the reference
            n = kp0.shape[0]
            if self._pgo_done is not None:
                back.wait_event(self._pgo_done)   # self.pose of the previous frame is produced on the PGO stream
            if n == 0:
                return FrameResult(self.pose, None, None, kp0, None)

            tr = ops.kp_track(kp0, maps1.flow, maps1.flow_cov, maps0, maps1, c.edgewidth, c.match_cov_default)
            pos0_Tc, pos_Tw, rot = ops.backproject(tr.kp0_uv, tr.vals[0], cam.K4, self.pose, want_rot=True)
            cov0, cov0_w, cov1 = ops.match_cov_pair(maps0.depth, tr.kp0_uv, tr.sigma0, maps1.depth, tr.kp1_uv, tr.sigma1,
                                                    *cam.K4, rot=rot, kernel_size=c.cov_kernel_size,
                                                    min_flow_cov=c.min_flow_cov, min_depth_cov=c.min_depth_cov)
            valid, n_valid = ops.obs_filter(tr.inbound, cov0, cov1, tr.vals, c.filters, c.filter_min_depth, self._max_depth)

            batch = ops.PGOBatch(
                offsets=self._offs[n], init_pose=self.pose.reshape(1, 7), intrinsics=self._intr, baseline=self._bl,
                pos_Tw=pos_Tw, pixel2_uv=tr.kp1_uv, cov_Tw=cov0_w, pixel2_d=tr.vals[4], pixel2_disp=tr.vals[5],
                pixel2_disp_cov=tr.vals[6], pixel2_uv_cov=tr.sigma1, obs2_covTc=cov1, valid=valid)
            ready = torch.cuda.Event()
            ready.record(back)
        side.wait_event(ready)
        with torch.cuda.stream(side):
            new_pose = torch.empty((1, 7), dtype=torch.float32, device=self.dev)
            pose64, info = ops.pgo_solve(batch, c.graph_type, self.lm, min_points=c.min_num_point, out_pose_f32=new_pose)
            if pose_sink is not None:
                pose_sink.copy_(new_pose.reshape(7), non_blocking=True)
            done = torch.cuda.Event()
            done.record(side)
        map_pts = None
        if c.mapping:
            # the reference maps only when tracking succeeded (:303-307) and then draws its second randperm of the frame
            ready.synchronize()
            if int(n_valid.item()) >= c.min_num_point:
                pend.cands_m._n = int(pend.host_count_m[0])
                with torch.cuda.stream(back):
                    muv = pend.cands_m.finish(c.map_num_point)
                    map_pts = ops.map_points(muv, maps0.depth, maps0.depth_cov, cam.K4, batch.init_pose, image=pend.image0,
                                             match_cov_default=c.match_cov_default, kernel_size=c.cov_kernel_size,
                                             min_flow_cov=c.min_flow_cov, min_depth_cov=c.min_depth_cov)
                    self._map_done = torch.cuda.Event()
                    self._map_done.record(back)
        self._pgo_done = done
        self._pgo_keep = (self._pgo_keep[1] if self._pgo_keep else None, (batch, tr, cov0, cov1, maps0, maps1, kp0, pos0_Tc, cands, pend))  # keep 2 frames of cross-stream tensors alive
        self.pose = new_pose.reshape(7)
        res = FrameResult(self.pose, pose64, info, kp0, n_valid)
        res.map_points = map_pts
        if self.keep_extras:
            res.extras = dict(tracked=tr, cov0=cov0, cov0_w=cov0_w, cov1=cov1, valid=valid, pos_Tw=pos_Tw,
                              maps1=maps1, cands=cands)
        return res

    def step(self, x: FrameInputs) -> FrameResult:
        """One ``run_pair`` start to finish (no cross-frame overlap); results are valid on the current stream."""
        assert self.maps_prev_for_next is not None, "call initialize() with the first frame"
        res = self.finish(self.enqueue_frontend(x))
        self.sync_pose()
        return res

    def sync_pose(self) -> None:
        """Make the current stream wait for the in-flight solve (needed before reading ``self.pose`` there)."""
        if self._pgo_done is not None:
            torch.cuda.current_stream().wait_event(self._pgo_done)
        if getattr(self, "_map_done", None) is not None:
            torch.cuda.current_stream().wait_event(self._map_done)

    def run(self, frames, pose_sink: torch.Tensor | None = None):
        """Software-pipelined stream: frame t+1's frontend is enqueued before frame t's host-side randperm, so the GPU
        never idles on the selector's host round trip.  Yields a FrameResult per frame (same results as ``step``)."""
        it = iter(frames)
        try:
            nxt = self.enqueue_frontend(next(it))
        except StopIteration:
            return
        i = 0
        while nxt is not None:
            cur = nxt
            try:
                nxt = self.enqueue_frontend(next(it))
            except StopIteration:
                nxt = None
            yield self.finish(cur, None if pose_sink is None else pose_sink[i])
            i += 1
        self.sync_pose()


@dataclass
class _Pending:
    maps0: "ops.FrontendMaps"
    maps1: "ops.FrontendMaps"
    cands: "ops.KeypointCandidates"
    host_count: torch.Tensor
    event: "torch.cuda.Event"
    cands_m: "ops.KeypointCandidates | None" = None     # mapping mode
    host_count_m: torch.Tensor | None = None
    image0: torch.Tensor | None = None


# ====================================================================================== native driver (default)
def stack_lanes(inputs: "list[FrameInputs]") -> FrameInputs:
    """Batch the per-sequence inputs of ``len(inputs)`` independent sequences ("lanes") along the pair axis — the
    reference's batching point (``Frontend.py:219-224``): lane l contributes pairs 2l (stereo) and 2l + 1 (temporal)."""
    # the per-lane inputs may have been produced on other streams: the concatenation below runs on the current one
    for x in inputs:
        if x.ready is not None:
            torch.cuda.current_stream().wait_event(x.ready)
    # one device map per pipe (lanes == 1), so per-lane images have no consumer in a batched step: refuse them rather than
    # drop them silently; the lanes advance in lock-step, so the step carries lane 0's timestamp
    assert len(inputs) == 1 or all(x.image is None for x in inputs), "stack_lanes: per-lane images are not carried"
    cat = lambda name, dim=0: (None if getattr(inputs[0], name) is None  # noqa: E731
                               else torch.cat([getattr(x, name) for x in inputs], dim=dim).contiguous())
    return FrameInputs(fmap1=cat("fmap1"), fmap2=cat("fmap2"), coords=cat("coords", 1), flow=cat("flow"), logcov=cat("logcov"),
                       flow8=cat("flow8"), cov8=cat("cov8"), up_mask=cat("up_mask"), cov_mask=cat("cov_mask"),
                       image=inputs[0].image if len(inputs) == 1 else None, time_ns=inputs[0].time_ns,
                       static=all(x.static for x in inputs))


class _NativeResult:
    """Result of one natively driven frame of one lane.  Every tensor is a VIEW into the pipe's arena: valid until two
    more frames have been finished (slots rotate); clone what must live longer."""

    def __init__(self, hp: "NativeHotPath", lane: int, n_sel: "int | None", n_cand: "int | None"):
        self._hp, self.lane, self._n_sel, self._n_cand = hp, lane, n_sel, n_cand
        self._fin = hp._n_fin          # finish counter at creation: views are resolved against it
        self._extras: "dict | None" = None
        self.map_points = None          # mapping mode: ops.MapPoints views of the frame's dense map points

    # Device-driven frames (round 6): the counts never reach the host on their own — the first access reads them back from the frame's backend slot
    # (blocks until that frame's front launch has run; off the hot path).
    def _resolve(self) -> None:
        if self._n_sel is None:
            self._n_cand, self._n_sel = self._hp._finished_counts(self._fin, self._age(), self.lane)

    @property
    def n_sel(self) -> int:
        self._resolve()
        return self._n_sel

    @property
    def n_cand(self) -> int:
        self._resolve()
        return self._n_cand

    @property
    def extras(self) -> dict:
        if self._extras is None:
            self._extras = self._hp._extras_of(self) if self._hp.keep_extras else {}
        return self._extras

    def _age(self) -> int:
        age = self._hp._n_fin - self._fin
        if age > 1:
            raise ops.L.MacvoHipError("this frame's buffers were recycled (results are views; clone them earlier)")
        return age

    def _rows(self, name, dtype, tail=()):
        """[n_sel, *tail] live rows of this lane in a per-keypoint table [lanes, cap, *tail]."""
        hp = self._hp
        return hp._view(name, self._age(), dtype, (hp.lanes, hp._cap) + tuple(tail))[self.lane, : self.n_sel]

    def _per_lane(self, name, dtype, tail):
        hp = self._hp
        return hp._view(name, self._age(), dtype, (hp.lanes,) + tuple(tail))[self.lane]

    @property
    def pose(self):
        """fp32 [7]: this lane's optimised pose (valid on a stream after ``sync_pose``)."""
        return self._per_lane("POSE", torch.float32, (7,))

    @property
    def kp0_uv(self):
        return self._rows("KP0", torch.int64, (2,))

    @property
    def n_valid(self):
        return self._per_lane("NVALID", torch.int32, ())[None] if self.n_sel else None

    @property
    def pose_f64(self):
        return self._per_lane("POSE64", torch.float64, (7,))[None] if self.n_sel else None

    @property
    def info(self):
        return self._per_lane("INFO", torch.float64, (4,))[None] if self.n_sel else None


class NativeHotPath:
    """Same contract as :class:`HotPath`, but the per-frame sequencing (streams, events, buffer rotation, ~30 launches)
    runs in C++ (``mv_frame_pipe_*``, csrc/frame_pipe.hip): two host calls per frame instead of ~30 Python-level ones.
    The Python loop was interpreter-bound (~370 us/frame, more than the GPU work); kernels, launch order and arguments
    are identical, so results are bit-identical to :class:`HotPath` (tests/test_gpu_native.py).

    ``lanes`` > 1 (BASELINE configs[4], "batch-32 frames per GPU"): that many INDEPENDENT sequences advance in lock-step
    through the same launches — one volume GEMM over ``2 * lanes`` pairs, lane-batched lookups / epilogue / selector /
    backend kernels, one batched LM solve.  Inputs are the per-lane :class:`FrameInputs` concatenated along the pair axis
    (:func:`stack_lanes`); ``finish`` then returns one result per lane.  Each lane draws its keypoint permutation from its
    own CPU generator (``generators[l]``; ``None`` = torch's global generator, which is what the reference consumes), in
    lane order — a lane seeded like a stand-alone run therefore selects exactly the keypoints of that stand-alone run."""

    def __init__(self, cam: Camera, cfg: HotPathConfig | None = None, device: str | torch.device = "cuda",
                 keep_extras: bool = False, lanes: int = 1, generators: "list | None" = None):
        self.cam, self.cfg = cam, cfg or HotPathConfig()
        if self.cfg.mapping and lanes != 1:
            raise ops.L.MacvoHipError("the dense-mapping tail (mapping=True) runs one sequence per pipe (lanes == 1), as the reference does")
        if self.cfg.use_graphs:
            raise ops.L.MacvoHipError("use_graphs belongs to the Python-sequenced pipeline.HotPath")
        if not 1 <= lanes <= ops.L.MV_MAX_LANES:
            raise ops.L.MacvoHipError(f"lanes must be in [1, {ops.L.MV_MAX_LANES}]")
        self.dev = torch.device(device)
        self.keep_extras = keep_extras
        self.lanes = int(lanes)
        self.generators = list(generators) if generators is not None else [None] * self.lanes
        assert len(self.generators) == self.lanes
        # integer entries = seeds of NATIVE per-lane generators (mv_frame_pipe_seed_lanes: MT19937 + Fisher-Yates in C++, the bits of
        # torch.Generator().manual_seed(seed) + torch.randperm at ~1/20 of the host time — what many-lane steps need)
        self._native_seeds = all(isinstance(g, int) and not isinstance(g, bool) for g in self.generators)
        if not self._native_seeds and any(isinstance(g, int) for g in self.generators):
            raise ops.L.MacvoHipError("generators: either all torch.Generator / None or all integer seeds")
        if self._native_seeds and self.cfg.mapping:
            raise ops.L.MacvoHipError("mapping=True draws its second permutation on the Python side: use torch generators")
        self._prev_image = None          # mapping: LEFT image of the previously enqueued frame (map-point colours, MACVO.py:326-328)
        self._images: list = []
        self._cap = max(self.cfg.num_point, 1)
        self._volume_ahead = os.environ.get("MV_PIPE_VOLUME_AHEAD", "1") != "0"   # A/B knobs of run()
        # frames in flight: never more than the slot rotation the library was built with (MV_MAX_PENDING, 3 in the stock build)
        # The default follows the stream layout the driver picks (mv_frame_pipe_default_depth): 3 for the round-5 layout of one- and two-lane pipes (even /
        # odd frames' decoder sides on two streams: a third frame in flight keeps both fed) and for batched pipes, 2 for the classic one-lane layout
        # ([r4] there 3 bought nothing and cost a period of latency, profiles/r04_latency_ab.log)
        lib = ops.L.load()
        self._depth = max(1, min(int(lib.mv_frame_pipe_max_pending()), int(os.environ.get("MV_PIPE_MAX_DEPTH", "3")),
                                 int(os.environ.get("MV_PIPE_DEPTH", str(lib.mv_frame_pipe_default_depth(self.lanes, int(bool(self.cfg.mapping))))))))
        self.lm = ops.lm_default_params()
        self._pipe = None
        self._arena = None
        self._views: dict = {}
        self._n_fin = 0
        self._n_enq = 0
        self._pending: list = []
        self._init_pose = None
        self._ncand = (ops.C.c_int32 * self.lanes)()
        self._nsel = (ops.C.c_int32 * self.lanes)()
        self._perm = torch.zeros((self.lanes, self._cap), dtype=torch.int64)
        self._ptr = ops.C.c_void_p()
        self._cnt = ops.C.c_size_t()

    # ------------------------------------------------------------------ construction (needs the first input's shapes)
    def _create(self, x: FrameInputs) -> None:
        L, C = ops.L, ops.C
        c, cam = self.cfg, self.cam
        lib = L.load()
        hwc = c.feature_layout == "hwc"
        pairs = x.fmap1.shape[0]
        if pairs != 2 * self.lanes:
            raise L.MacvoHipError(f"inputs carry {pairs} pairs but the pipe has {self.lanes} lane(s) (2 pairs per lane)")
        chans = x.fmap1.shape[-1] if hwc else x.fmap1.shape[1]
        dt = {torch.float32: L.MV_F32, torch.float16: L.MV_F16, torch.bfloat16: L.MV_BF16}[x.fmap1.dtype]
        bl_fx = float(cam.baseline) * float(cam.fx)
        max_depth = cam.fx * cam.baseline if c.max_depth == "auto" else float(c.max_depth)
        pc = L.mvFramePipeConfig(
            H=cam.H, W=cam.W, C=chans, pairs=pairs, iters=x.coords.shape[0], radius=c.radius, feat_dtype=dt,
            layout=L.MV_LAYOUT_HWC if hwc else L.MV_LAYOUT_CHW, volume_split={"exact": 0, "split3": 3, "split2": 2, "bf16x3": L.MV_PACK_BF16X3, "f16x2": L.MV_PACK_F16X2}[c.volume_precision] if dt == L.MV_F32 else (L.MV_VOL_ENC16 if (c.volume_store == "encoder" and dt == L.MV_F16 and c.radius == 4) else 0),   # 16-bit features: one kernel family
            selector_mode=L.MV_KP_NODEPTH if c.selector == "nodepth" else L.MV_KP_FULL,
            kp_kernel_size=c.kp_kernel_size, kp_mask_width=c.kp_mask_width, num_point=c.num_point, edgewidth=c.edgewidth,
            min_num_point=c.min_num_point, graph_type=ops._GRAPH[c.graph_type], filters=c.filters,
            cov_kernel_size=c.cov_kernel_size, fx=cam.fx, fy=cam.fy, cx=cam.cx, cy=cam.cy, baseline=cam.baseline,
            bl_fx=bl_fx, bl_fx_sq=bl_fx ** 2, match_cov_default=c.match_cov_default, max_match_cov=c.max_match_cov,
            max_depth_cov=c.max_depth_cov, max_depth=max_depth, min_flow_cov_sq=c.min_flow_cov ** 2,
            min_depth_cov=c.min_depth_cov, filter_min_depth=c.filter_min_depth, mapping=int(c.mapping), map_num_point=c.map_num_point,
            map_mask_width=c.map_mask_width, async_backend=0 if c.async_backend is None else (1 if c.async_backend else -1), map_max_depth=c.map_max_depth, map_max_depth_cov=c.map_max_depth_cov, lm=self.lm)
        nbytes = lib.mv_frame_pipe_arena_bytes(C.byref(pc))
        if nbytes == 0:
            raise L.MacvoHipError("mv_frame_pipe_arena_bytes: invalid configuration")
        self._arena = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.dev)
        self._base = (self._arena.data_ptr() + 255) & ~255
        pipe = C.c_void_p()
        L.check(lib.mv_frame_pipe_create(C.byref(pc), self._base, nbytes, C.byref(pipe)), "mv_frame_pipe_create")
        self._pipe, self._lib, self._pc = pipe, lib, pc
        self.device_driven = False
        if self._native_seeds:
            seeds = (C.c_uint64 * self.lanes)(*[int(g) & 0xFFFFFFFFFFFFFFFF for g in self.generators])
            L.check(lib.mv_frame_pipe_seed_lanes(pipe, seeds), "mv_frame_pipe_seed_lanes")
            # round 6: the generators also live in device memory and — wherever it applies — the frame is device-driven: the permutation head is drawn
            # inside the backend's front launch, `finish` never waits for the GPU (csrc/frame_pipe.hip, csrc/randperm_dev.h; MV_PIPE_DEVICE_DRAW=0 restores
            # the host draw).  Same keypoints, same poses (tests/test_gpu_lanes.py).
            self.device_driven = bool(lib.mv_frame_pipe_device_draw(pipe))
        # the volume buffers hold every query's slice in 4 x 4-cell tiles (Fast-mode pipes; MV_PIPE_TILED): read them with corr_lookup(tiled=True)
        self.volume_tiled = bool(lib.mv_frame_pipe_volume_tiled(pipe))
        self.host_threads = int(lib.mv_frame_pipe_host_threads(pipe))      # 1, or 2 with the backend launch thread
        self._counts_cache: dict = {}
        self.host_issue_s = self.host_wait_s = 0.0
        self.host_frames = 0
        if self._init_pose is not None:
            self._set_pose(self._init_pose)

    def close(self) -> None:
        """Destroy the native pipe NOW (drains its streams, frees its HIP streams / events; the arena goes back to torch's allocator).  Result objects keep a
        reference to their pipe, so dropping the last NAME of a pipe does not destroy it while any result of it is alive — and a pipe that lingers keeps its four
        HIP streams: the next pipe's streams then share hardware queues with them (measured: a 32-lane pipe at 6.3 k instead of 7.8 k frames/s behind a lingering
        one-lane pipe, profiles/probes/r6_queue_history.py).  Views handed out earlier become invalid."""
        if getattr(self, "_pipe", None):
            self._lib.mv_frame_pipe_destroy(self._pipe)
            self._pipe = None
            self._views.clear()
            self._arena = None

    def __del__(self):
        self.close()

    def _set_pose(self, pose: torch.Tensor) -> None:
        host = pose.detach().to("cpu", torch.float32).reshape(-1, 7)
        if host.shape[0] == 1 and self.lanes > 1:
            host = host.expand(self.lanes, 7)
        assert host.shape[0] == self.lanes, "pose must be [7] or [lanes, 7]"
        host = host.contiguous()
        ops.L.check(self._lib.mv_frame_pipe_set_pose(self._pipe, host.data_ptr()), "mv_frame_pipe_set_pose")

    def _view(self, name: str, age: int, dtype: torch.dtype, shape: tuple) -> torch.Tensor:
        if self._pipe is None:
            raise ops.L.MacvoHipError("this pipe has been closed (or was never created): its buffers are gone")
        ops.L.check(self._lib.mv_frame_pipe_buffer(self._pipe, ops.L.FB[name], age, ops.C.byref(self._ptr),
                                                   ops.C.byref(self._cnt)), f"mv_frame_pipe_buffer({name})")
        key = (self._ptr.value, dtype, shape)
        v = self._views.get(key)
        if v is None:
            off = self._ptr.value - self._arena.data_ptr()
            n = 1
            for s in shape:
                n *= s
            assert n <= self._cnt.value or n == 0, (name, shape, self._cnt.value)
            v = self._arena[off: off + n * torch.empty((), dtype=dtype).element_size()].view(dtype).view(shape)
            if len(self._views) > 256:
                self._views.clear()
            self._views[key] = v
        return v

    @property
    def pose(self) -> torch.Tensor:
        """fp32 ``[7]`` (``[lanes, 7]`` for lanes > 1) view of the newest solve's output (valid on a stream after
        :meth:`sync_pose`)."""
        v = self._view("POSE", 0, torch.float32, (self.lanes, 7))
        return v[0] if self.lanes == 1 else v

    @pose.setter
    def pose(self, value: torch.Tensor) -> None:
        """Override the prior of the next frame (blocking; e.g. an external motion model)."""
        if self._pipe is None:
            self._init_pose = value
        else:
            self._set_pose(value)

    @property
    def last_tokens(self) -> torch.Tensor:
        """Window-lookup output of the newest frame's last decoder iteration ``[pairs, (2r+1)^2, H/8, W/8]``."""
        k = 2 * self.cfg.radius + 1
        return self._view("TOKENS", 0, torch.float32, (self._pc.pairs, k * k, self.cam.H // 8, self.cam.W // 8))

    def maps(self, age: int = 0, lane: int = 0) -> "ops.FrontendMaps":
        H, W = self.cam.H, self.cam.W
        v = lambda n, c: self._view(n, age, torch.float32, (self.lanes, c, H, W))[lane: lane + 1]  # noqa: E731
        return ops.FrontendMaps(v("DEPTH", 1), v("DEPTH_COV", 1), v("DISPARITY", 1), v("DISPARITY_COV", 1), None,
                                v("MATCH_FLOW", 2), v("MATCH_COV", 3))

    # ------------------------------------------------------------------ frame API
    def _inputs(self, x: FrameInputs):
        st = getattr(x, "_native_struct", None)
        if st is not None and x.static:
            return st
        q = lambda t, dt=torch.float32: None if t is None else ops._req(t, dt, "frame input").data_ptr()  # noqa: E731
        st = ops.L.mvFrameInputs(q(x.fmap1, x.fmap1.dtype), q(x.fmap2, x.fmap2.dtype), q(x.coords), q(x.flow), q(x.logcov),
                                 q(x.flow8), q(x.cov8), q(x.up_mask), q(x.cov_mask))
        if x.flow8 is not None:
            st.flow, st.logcov = None, None
        x._native_struct = st
        return st

    def _enqueue(self, x: FrameInputs, with_selector: bool) -> None:
        if self._pipe is None:
            self._create(x)
        if x.ready is not None:
            torch.cuda.current_stream().wait_event(x.ready)
        ops.L.check(self._lib.mv_frame_pipe_enqueue(self._pipe, ops.C.byref(self._inputs(x)), ops._stream(),
                                                    int(with_selector)), "mv_frame_pipe_enqueue")
        self._n_enq += 1

    def attach_map(self, devmap, K: torch.Tensor, T_BS: torch.Tensor | None = None) -> None:
        """Register every finished frame in a :class:`macvo_amd.devmap.DeviceVisualMap`: the frame's
        tables go from the tracking kernels into the map's SoA stores on the pipe's own streams — no ``.cpu()`` round trip
        (the reference: Odometry/MACVO.py:235-266, ~25 device-to-host copies per frame).  lanes == 1.  Call before
        :meth:`initialize`."""
        if self.lanes != 1:
            raise ops.L.MacvoHipError("attach_map: one map per pipe, lanes must be 1")
        self._map = devmap
        self._map_K = K.to(self.dev, torch.float32).reshape(3, 3).contiguous()
        self._map_TBS = (torch.tensor([0, 0, 0, 0, 0, 0, 1.0]) if T_BS is None else T_BS).to(self.dev, torch.float32).reshape(7).contiguous()
        self._times: list = []

    def initialize(self, x: FrameInputs, init_pose: torch.Tensor | None = None) -> None:
        """Frame 0: ``MACVO.initialize`` (:158-171) — depth only, pose = prior."""
        self._init_pose = init_pose
        self._enqueue(x, False)
        self._prev_image = x.image
        if getattr(self, "_map", None) is not None:   # MACVO.initialize pushes the first frame at the prior (:162-169)
            self._map.push_frame(K=self._map_K, T_BS=self._map_TBS, baseline=self.cam.baseline, time_ns=x.time_ns, prior_pose=init_pose)
            torch.cuda.current_stream().synchronize()   # later frames are appended on the pipe's streams: order them after this one

    @traced("Frontend.estimate")
    def enqueue_frontend(self, x: FrameInputs):
        assert self._n_enq >= 1, "call initialize() with the first frame"
        self._enqueue(x, True)
        if self.cfg.mapping:
            self._images.append(self._prev_image)
            self._prev_image = x.image
        if getattr(self, "_map", None) is not None:
            self._times.append(int(x.time_ns))
        return x

    def enqueue_volume(self, x: FrameInputs) -> None:
        """Issue only the cost-volume GEMM of the frame the NEXT :meth:`enqueue_frontend` call will complete (same ``x``).
        The GEMM needs nothing but the feature maps and a free volume buffer, so it can be queued a frame ahead — before the
        host blocks on the previous frame's candidate count — and the GEMM stream never waits for the host."""
        if x.ready is not None:
            torch.cuda.current_stream().wait_event(x.ready)
        ops.L.check(self._lib.mv_frame_pipe_enqueue_volume(self._pipe, ops.C.byref(self._inputs(x)), ops._stream()),
                    "mv_frame_pipe_enqueue_volume")

    @traced("Odom_Runtime")
    def finish(self, pend=None, pose_sink: torch.Tensor | None = None):
        """Host half of a frame: wait for the candidate counts, draw the permutations (CPU generators, lane order), enqueue
        the pose-dependent kernels.  Returns a :class:`_NativeResult` (a list of them, one per lane, for lanes > 1)."""
        L, lib = ops.L, self._lib
        if self.device_driven:
            L.check(lib.mv_frame_pipe_release(self._pipe, ops._stream()), "mv_frame_pipe_release")
            L.check(lib.mv_frame_pipe_finish_device(self._pipe, None if pose_sink is None else pose_sink.data_ptr()), "mv_frame_pipe_finish_device")
            return self._finished()
        if self._native_seeds:
            L.check(lib.mv_frame_pipe_release(self._pipe, ops._stream()), "mv_frame_pipe_release")
            L.check(lib.mv_frame_pipe_finish_seeded(self._pipe, None if pose_sink is None else pose_sink.data_ptr(), self._ncand, self._nsel),
                    "mv_frame_pipe_finish_seeded")
            return self._finished()
        L.check(lib.mv_frame_pipe_wait_candidates(self._pipe, self._ncand), "mv_frame_pipe_wait_candidates")
        num = self.cfg.num_point
        if self.lanes == 1:
            n = self._ncand[0]
            g = self.generators[0]
            # global CPU generator by default, exactly as the reference (KeypointSelector.py:331,404)
            perm = (torch.randperm(n) if g is None else torch.randperm(n, generator=g))[:num]
            self._nsel[0] = perm.numel()
            perm_ptr = perm.data_ptr() if perm.numel() else None
        else:
            for l in range(self.lanes):
                n = self._ncand[l]
                g = self.generators[l]
                perm = (torch.randperm(n) if g is None else torch.randperm(n, generator=g))[:num]
                k = perm.numel()
                self._nsel[l] = k
                if k:
                    self._perm[l, :k] = perm
            perm_ptr = self._perm.data_ptr()
        # views of earlier results may still be being read on the caller's stream: the pipe recycles their buffers behind that
        L.check(lib.mv_frame_pipe_release(self._pipe, ops._stream()), "mv_frame_pipe_release")
        L.check(lib.mv_frame_pipe_finish(self._pipe, perm_ptr, self._nsel, None if pose_sink is None else pose_sink.data_ptr()),
                "mv_frame_pipe_finish")
        return self._finished()

    def _map_tail(self, mp):
        """Dense-mapping tail of the frame just finished (Odometry/MACVO.py:303-337): the reference maps only when tracking
        succeeded — so the frame's observation count has to reach the host first (one blocking wait per frame: mapping mode gives
        up the driver's run-ahead, as the reference's own `.cpu()` calls do) — and then draws its SECOND randperm of the frame from
        the same CPU generator."""
        L, lib, c = ops.L, self._lib, self.cfg
        image0 = self._images.pop(0)
        nv, nm = ops.C.c_int32(0), ops.C.c_int32(0)
        L.check(lib.mv_frame_pipe_wait_tracked(self._pipe, ops.C.byref(nv), ops.C.byref(nm)), "mv_frame_pipe_wait_tracked")
        if nv.value < c.min_num_point:
            return None
        g = self.generators[0]
        perm = (torch.randperm(nm.value) if g is None else torch.randperm(nm.value, generator=g))[: c.map_num_point]
        n = perm.numel()
        img = None if image0 is None else ops._req(image0.reshape(3, self.cam.H, self.cam.W), torch.float32, "image")
        if mp is not None:
            if mp.map_rows_upper + n >= mp.cap["map_points"]:
                self.synchronize()
                mp.reserve_map_points(n)
                torch.cuda.synchronize()
            mp.map_rows_upper += n
        L.check(lib.mv_frame_pipe_map_points(self._pipe, perm.data_ptr() if n else None, n, None if img is None else img.data_ptr(),
                                             None if mp is None else ops.C.byref(mp.stores())), "mv_frame_pipe_map_points")
        self._map_keep = img
        f32 = torch.float32
        v = lambda name, dt, tail: self._view(name, 0, dt, (n,) + tail)  # noqa: E731
        return ops.MapPoints(v("MAP_UV", f32, (2,)), v("MAP_D", f32, ()), v("MAP_SDD", f32, ()), v("MAP_TC", f32, (3,)), v("MAP_TW", f32, (3,)),
                             v("MAP_COV", torch.float64, (3, 3)), None if img is None else v("MAP_COLOR", torch.uint8, (3,)))

    def _finished(self):
        """Bookkeeping behind a finish call: map registration, result views."""
        L, lib = ops.L, self._lib
        self._n_fin += 1
        mp = getattr(self, "_map", None)
        dd = self.device_driven
        if mp is not None:
            n_rows = self._cap if dd else int(self._nsel[0])   # (device-driven: an upper bound; the append compacts by the `valid` mask)
            if mp.n_frames + 1 >= mp.cap["frames"] or mp.rows_upper + n_rows >= mp.cap["match"]:
                self.synchronize()                          # growth re-allocates the stores: rare (capacity doubles), so simply drain
                mp.reserve(n_rows)
                torch.cuda.synchronize()
            L.check(lib.mv_frame_pipe_map_append(self._pipe, ops.C.byref(mp.stores()), mp.n_frames, mp.n_frames - 1,
                                                 self._map_K.data_ptr(), self._map_TBS.data_ptr(), float(self.cam.baseline),
                                                 self._times.pop(0), None), "mv_frame_pipe_map_append")
            mp.n_frames += 1
            mp.rows_upper += n_rows
        map_pts = self._map_tail(mp) if self.cfg.mapping else None
        out = []
        for l in range(self.lanes):
            res = _NativeResult(self, l, None if dd else self._nsel[l], None if dd else self._ncand[l])
            res.map_points = map_pts
            if self.keep_extras and not dd:
                res.extras   # noqa: B018  (host-driven frames: build the views now, as before)
            out.append(res)
        return out[0] if self.lanes == 1 else out

    def _extras_of(self, res: "_NativeResult") -> dict:
        n_sel, l = res.n_sel, res.lane
        if not n_sel:
            return {}
        f32, f64 = torch.float32, torch.float64
        vals = self._view("VALS", res._age(), f32, (11, self.lanes, self._cap))[:, l, :n_sel]
        tr = ops.TrackedKeypoints(res._rows("KP0F", f32, (2,)), res._rows("KP1", f32, (2,)),
                                  res._rows("INBOUND", torch.bool), vals,
                                  res._rows("SIGMA0", f32, (3,)), res._rows("SIGMA1", f32, (3,)))
        return dict(tracked=tr, cov0=res._rows("COV0", f64, (3, 3)), cov0_w=res._rows("COV0W", f64, (3, 3)),
                    cov1=res._rows("COV1", f64, (3, 3)), valid=res._rows("VALID", torch.bool),
                    pos_Tw=res._rows("POS_TW", f32, (3,)), n_cand=res.n_cand)

    def _finished_counts(self, fin: int, age: int, lane: int):
        """(n_cand, n_sel) of lane ``lane`` of the frame that was finish number ``fin`` (device-driven frames; blocking read-back)."""
        got = self._counts_cache.get(fin)
        if got is None:
            nc, ns = (ops.C.c_int32 * self.lanes)(), (ops.C.c_int32 * self.lanes)()
            ops.L.check(self._lib.mv_frame_pipe_finished_counts(self._pipe, age, nc, ns), "mv_frame_pipe_finished_counts")
            got = (list(nc), list(ns))
            if len(self._counts_cache) > 64:
                self._counts_cache.clear()
            self._counts_cache[fin] = got
        return got[0][lane], got[1][lane]

    def step(self, x: FrameInputs):
        """One ``run_pair`` start to finish (no cross-frame overlap); results are valid on the current stream."""
        self.enqueue_frontend(x)
        res = self.finish()
        self.sync_all()
        return res

    def sync_pose(self) -> None:
        """Make the current stream wait for the newest FINISHED frame's backend and solve (its results: keypoints, covariances,
        pose) — not for the frontends of the frames :meth:`run` has already queued behind it.  Read a result right after this
        call: its buffers are recycled two finishes later."""
        if self._pipe is not None:
            ops.L.check(self._lib.mv_frame_pipe_sync(self._pipe, ops._stream(), 2), "mv_frame_pipe_sync")

    def sync_all(self) -> None:
        """Make the current stream wait for everything the pipe has enqueued (frontends of queued frames included)."""
        if self._pipe is not None:
            ops.L.check(self._lib.mv_frame_pipe_sync(self._pipe, ops._stream(), 0), "mv_frame_pipe_sync")

    def time_volume(self, max_launches: int) -> None:
        """Record a HIP-event pair around each of the next ``max_launches`` volume GEMMs (on the stream they run on)."""
        ops.L.check(self._lib.mv_frame_pipe_time_volume(self._pipe, int(max_launches)), "mv_frame_pipe_time_volume")

    def time_detail(self, on: bool) -> None:
        """``False``: timed frames record only the event pair around their GEMM (no timeline events on the other streams)."""
        ops.L.check(self._lib.mv_frame_pipe_time_detail(self._pipe, int(bool(on))), "mv_frame_pipe_time_detail")

    def volume_times_ms(self) -> list:
        cap = self._pc_timed = 1 << 16
        buf = (ops.C.c_float * cap)()
        n = ops.C.c_int(0)
        ops.L.check(self._lib.mv_frame_pipe_volume_times(self._pipe, buf, cap, ops.C.byref(n)), "mv_frame_pipe_volume_times")
        return list(buf[: n.value])

    def volume_starts_ms(self) -> list:
        """Start of each timed GEMM, ms since the first timed one."""
        cap = 1 << 16
        buf = (ops.C.c_float * cap)()
        n = ops.C.c_int(0)
        ops.L.check(self._lib.mv_frame_pipe_volume_starts(self._pipe, buf, cap, ops.C.byref(n)), "mv_frame_pipe_volume_starts")
        return list(buf[: n.value])

    def timeline_ms(self) -> list:
        """[(GEMM start, GEMM end, last lookup done, selector done)] per timed frame, ms since the first timed GEMM start."""
        cap = 1 << 14
        buf = (ops.C.c_float * (4 * cap))()
        n = ops.C.c_int(0)
        ops.L.check(self._lib.mv_frame_pipe_timeline(self._pipe, buf, cap, ops.C.byref(n)), "mv_frame_pipe_timeline")
        return [tuple(buf[4 * i: 4 * i + 4]) for i in range(n.value)]

    def timeline_backend_ms(self) -> list:
        """[(backend start, backend end, pose_apply start, solve end)] per timed frame, same time base as :meth:`timeline_ms`."""
        cap = 1 << 14
        buf = (ops.C.c_float * (4 * cap))()
        n = ops.C.c_int(0)
        ops.L.check(self._lib.mv_frame_pipe_timeline_backend(self._pipe, buf, cap, ops.C.byref(n)), "mv_frame_pipe_timeline_backend")
        return [tuple(buf[4 * i: 4 * i + 4]) for i in range(n.value)]

    def synchronize(self) -> None:
        if self._pipe is not None:
            ops.L.check(self._lib.mv_frame_pipe_sync(self._pipe, None, 1), "mv_frame_pipe_sync")

    def run(self, frames, pose_sink: torch.Tensor | None = None, depth: int | None = None):
        """Software-pipelined stream: up to ``depth`` (default 3 = the pipe's slot rotation) tracked frames are enqueued ahead
        of the frame whose candidate count the host waits for, plus the volume GEMM of the one after — the GPU-side chain
        volume -> 12 lookups -> selector of a frame takes ~2.5 frame periods when it shares the chip with the next GEMMs, so
        the host has to run that far ahead for the GEMM stream to stay busy.  ``pose_sink``: ``[steps, 7]``
        (``[steps, lanes, 7]`` for lanes > 1) device tensor receiving each step's poses."""
        depth = self._depth if depth is None else depth
        it = iter(frames)
        nxt = next(it, None)
        if self._pipe is not None and self.device_driven:
            # Device-driven frames: nothing in a frame waits for the host, so a frame is enqueued and finished in one go (the next frame's GEMM in between, as
            # in the host-driven order); how far the host runs ahead of the GPU is bounded by the HIP queues, the slot rotation is guarded by events.
            i = 0
            # ... and by `lag`: the host stays at most that many finished frames ahead of the GPU's front launches (flow control on an event two frames old, not a
            # wait on the critical chain).  Measured at 640x480 (profiles/r06_device_draw_ab.log): lag 0 / 1 / 2 / 3 / 4 / 6 / unbounded = 4.13 / 6.01 / 6.51 / 6.46 / 6.34 /
            # 6.24 / 4.07 k frames/s — with hundreds of frames queued the same kernels take 1.6x as long (deep queues of cross-queue barriers), so the default is 2.
            # (Also measured and dropped: flow control on the SELECTOR's event instead of the front launch's — 30 us earlier, i.e. deeper: 5.0 k instead of 5.4 k on
            # the 20-step line — and the host-drawn frame's order, a frame finished only once its selector is done: 5.0 k / 5.9-6.4 k at 300 steps.)
            lag = min(int(os.environ.get("MV_PIPE_DD_AHEAD", "2")), 6)      # (the driver keeps a ring of 8 front-launch events)
            clock = time.perf_counter
            while nxt is not None:
                t0 = clock()
                self.enqueue_frontend(nxt)
                nxt = next(it, None)
                if self._volume_ahead and nxt is not None:
                    self.enqueue_volume(nxt)
                res = self.finish(None, None if pose_sink is None else pose_sink[i])
                t1 = clock()
                i += 1
                if lag >= 0:
                    ops.L.check(self._lib.mv_frame_pipe_wait_finished(self._pipe, lag), "mv_frame_pipe_wait_finished")
                # host accounting (bench.py `host_us_per_frame`): time spent issuing a frame vs time spent in the flow-control wait (= ahead of the GPU)
                self.host_issue_s += t1 - t0
                self.host_wait_s += clock() - t1
                self.host_frames += 1
                yield res
            self.sync_all()
            return
        state = {"nxt": nxt, "vol": False, "pending": 0}

        def pump():
            while state["nxt"] is not None and state["pending"] < depth:
                self.enqueue_frontend(state["nxt"])       # completes the frame (its GEMM may already be queued)
                state["pending"] += 1
                state["nxt"], state["vol"] = next(it, None), False
            if self._volume_ahead and state["nxt"] is not None and not state["vol"]:
                self.enqueue_volume(state["nxt"])
                state["vol"] = True

        pump()
        i = 0
        while state["pending"]:
            res = self.finish(None, None if pose_sink is None else pose_sink[i])
            state["pending"] -= 1
            i += 1
            pump()                                        # refill before handing the result out: the GPU stays fed
            yield res
        self.sync_all()

    def _has_pending(self) -> bool:
        return self._n_fin < self._n_enq - 1          # frame 0 (initialize) is never finished
