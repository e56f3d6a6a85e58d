"""Device-resident ``VisualMap``: the tracking map of one sequence kept in HBM.

Mirrors ``Module/Map/VisualMap.py:15-133`` — the three SoA stores ``frames`` / ``points`` / ``match`` with the reference's
field names and dtypes (:23-69) and the edge tables ``frame2match`` / ``frame2map`` (``DenseEdge_Multi``), ``match2frame1``
/ ``match2frame2`` / ``match2point`` (``SingleEdge``), ``point2match`` (``SparseEdge_Multi``) of ``Module/Map/Graph.py`` —
as the reference's ``MACVO.run_pair`` fills them (``Odometry/MACVO.py:158-171,244-311``).  The reference keeps all of this
on the CPU and pays ~25 ``.cpu()`` copies per frame (:235-266); here a frame is registered by ONE HIP launch
(``mv_map_append``) that reads the tracking kernels' tables where they lie, compacts the kept rows in order
(``bundle[mask]``) and advances device-side counters — no host synchronisation, no D2H traffic until the map is written out.

Outputs (``Odometry/Interface.py:47-53``): :meth:`write` produces ``poses.npy`` (``[time_ns, body pose]`` rows) and
``tensor_map.npz`` with exactly the keys of ``VisualMap.serialize`` (:104-116 — note the reference's double slash in
``frames//K``: ``TensorBundle.serialize`` joins its ``"frames/"`` prefix with another ``/``).  :meth:`motion_interpolate`
is ``MotionInterpolate.elaborate_map`` (``Module/MapProcessor.py:52-76``) as one HIP launch.

PyTorch is used for device memory only (growth = allocate + copy, ``AutoScalingTensor._scale_up_to`` semantics,
``Utility/Extensions/TensorExtension.py:86-97``); every arithmetic / indexing step is in ``csrc/visual_map.hip``.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from . import _lib as L

C = L.C

_FRAME = {"K": ((3, 3), torch.float32), "baseline": ((), torch.float32), "pose": ((7,), torch.float32),
          "T_BS": ((7,), torch.float32), "need_interp": ((), torch.bool), "time_ns": ((), torch.int64)}
_POINT = {"pos_Tw": ((3,), torch.float32), "cov_Tw": ((3, 3), torch.float64), "color": ((3,), torch.uint8)}
_MATCH = {"pixel1_uv": ((2,), torch.float32), "pixel2_uv": ((2,), torch.float32), "pixel1_d": ((1,), torch.float32),
          "pixel2_d": ((1,), torch.float32), "pixel1_disp": ((1,), torch.float32), "pixel2_disp": ((1,), torch.float32),
          "pixel1_disp_cov": ((1,), torch.float32), "pixel2_disp_cov": ((1,), torch.float32),
          "obs1_covTc": ((3, 3), torch.float64), "obs2_covTc": ((3, 3), torch.float64),
          "pixel1_uv_cov": ((3,), torch.float32), "pixel2_uv_cov": ((3,), torch.float32),
          "pixel1_d_cov": ((1,), torch.float32), "pixel2_d_cov": ((1,), torch.float32)}


def _grow_to(size: int) -> int:
    return int(2 ** math.ceil(math.log2(size + 1)))          # AutoScalingTensor._scale_up_to (:87)


class DeviceVisualMap:
    def __init__(self, device: str | torch.device = "cuda", init_size: int = 1024, max_pt_obs: int = 5,
                 max_frame_range: int = 2):
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise L.MacvoHipError("DeviceVisualMap lives on the GPU (no CPU fallback)")
        self.lib = L.load()
        self.max_pt_obs, self.max_frame_range = max_pt_obs, max_frame_range
        self.cap = {"frames": init_size, "match": init_size, "points": init_size, "map_points": init_size}
        self.frames = {k: self._alloc(init_size, s, d) for k, (s, d) in _FRAME.items()}
        self.points = {k: self._alloc(init_size, s, d) for k, (s, d) in _POINT.items()}
        self.match = {k: self._alloc(init_size, s, d) for k, (s, d) in _MATCH.items()}
        self.map_points = {k: self._alloc(init_size, s, d) for k, (s, d) in _POINT.items()}     # VisualMap.map_points (VisualMap.py:47-54)
        i64 = torch.int64
        self.edges = {
            "frame2match_ranges": self._alloc(init_size, (max_frame_range, 2), i64, -1),
            "frame2match_num": self._alloc(init_size, (), i64, 0),
            "frame2map_ranges": self._alloc(init_size, (max_frame_range, 2), i64, -1),
            "frame2map_num": self._alloc(init_size, (), i64, 0),
            "match2frame1": self._alloc(init_size, (), i64, -1),
            "match2frame2": self._alloc(init_size, (), i64, -1),
            "match2point": self._alloc(init_size, (), i64, -1),
            "point2match_edges": self._alloc(init_size, (max_pt_obs,), i64, -1),
            "point2match_deg": self._alloc(init_size, (), i64, 0),
        }
        self.counts = torch.zeros(6, dtype=i64, device=self.dev)      # {frames, matches, points, lost frames, refused appends, map points}: advanced on device
        self.n_frames = 0                                             # exact (one per push)
        self.rows_upper = 0                                           # upper bound of matches == points pushed
        self.map_rows_upper = 0                                       # upper bound of map points pushed
        self._stores = None

    def _alloc(self, n, shape, dtype, fill=None):
        t = torch.empty((n,) + tuple(shape), dtype=dtype, device=self.dev)
        if fill is not None:
            t.fill_(fill)
        return t

    # ------------------------------------------------------------------ capacity (host-side upper bounds only)
    _EDGE_FILL = {"frame2match_ranges": -1, "frame2match_num": 0, "frame2map_ranges": -1, "frame2map_num": 0,
                  "match2frame1": -1, "match2frame2": -1, "match2point": -1, "point2match_edges": -1, "point2match_deg": 0}

    def _regrow(self, table: dict, keys, new_cap: int) -> None:
        for k in keys:
            t = table[k]
            nt = torch.empty((new_cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=self.dev)
            if k in self._EDGE_FILL:
                nt.fill_(self._EDGE_FILL[k])
            nt[: t.shape[0]].copy_(t)
            table[k] = nt
        self._stores = None

    def reserve(self, new_rows: int, new_map_points: int = 0) -> None:
        """Make room for one more frame with at most ``new_rows`` kept observations and ``new_map_points`` dense map points
        (``AutoScalingTensor.push`` :106-114 grows when size + n >= capacity; copies are enqueued on the current stream)."""
        need_mp = self.map_rows_upper + new_map_points
        if new_map_points and need_mp >= self.cap["map_points"]:
            cap = _grow_to(need_mp)
            self._regrow(self.map_points, list(self.map_points), cap)
            self.cap["map_points"] = cap
        need_f, need_r = self.n_frames + 1, self.rows_upper + new_rows
        if need_f >= self.cap["frames"]:
            cap = _grow_to(need_f)
            self._regrow(self.frames, list(self.frames), cap)
            self._regrow(self.edges, [k for k in self.edges if k.startswith("frame2")], cap)
            self.cap["frames"] = cap
        if need_r >= self.cap["match"]:
            cap = _grow_to(need_r)
            self._regrow(self.match, list(self.match), cap)
            self._regrow(self.points, list(self.points), cap)
            self._regrow(self.edges, [k for k in self.edges if k.startswith(("match2", "point2"))], cap)
            self.cap["match"] = self.cap["points"] = cap

    # ------------------------------------------------------------------ C-ABI view of the stores
    def stores(self) -> "L.mvMapStores":
        if self._stores is None:
            p = {k: v.data_ptr() for tbl in (self.frames, self.points, self.match, self.edges) for k, v in tbl.items()}
            mp = {f"mp_{k}": v.data_ptr() for k, v in self.map_points.items()}
            self._stores = L.mvMapStores(**p, **mp, counts=self.counts.data_ptr(), max_pt_obs=self.max_pt_obs,
                                         max_frame_range=self.max_frame_range, cap_frames=self.cap["frames"],
                                         cap_match=self.cap["match"], cap_points=self.cap["points"],
                                         cap_map_points=self.cap["map_points"])
        return self._stores

    def push_frame(self, *, K, T_BS, baseline: float, time_ns: int, prior_pose=None, tracked=None, valid=None, cov0=None,
                   cov1=None, pos_Tw=None, cov0_world=None, color=None, min_num_point: int = 10) -> int:
        """Register one frame from explicit tensors (``tracked`` = :class:`ops.TrackedKeypoints` with a contiguous ``[11, N]``
        table); the native frame driver uses ``mv_frame_pipe_map_append`` instead.  Returns the frame's map index."""
        from . import ops

        n = 0 if tracked is None else tracked.kp0_uv.shape[0]
        self.reserve(n)
        f32 = torch.float32
        Kd = ops._req(K.to(self.dev, f32).reshape(3, 3), f32, "K")
        Td = ops._req(T_BS.to(self.dev, f32).reshape(7), f32, "T_BS")
        pr = None if prior_pose is None else ops._req(prior_pose.to(self.dev, f32).reshape(7), f32, "prior_pose")
        q = lambda t, dt, nm: None if t is None else ops._req(t, dt, nm).data_ptr()  # noqa: E731
        fr = L.mvMapFrame(n_rows=n, table_stride=n if n else 0, prev_frame=self.n_frames - 1, min_num_point=min_num_point,
                          valid=q(None if valid is None else valid.view(torch.uint8) if valid.dtype == torch.bool else valid, torch.uint8, "valid"),
                          kp0=q(None if n == 0 else tracked.kp0_uv, f32, "kp0"), kp1=q(None if n == 0 else tracked.kp1_uv, f32, "kp1"),
                          vals=q(None if n == 0 else tracked.vals, f32, "vals"),
                          sigma0=q(None if n == 0 else tracked.sigma0, f32, "sigma0"), sigma1=q(None if n == 0 else tracked.sigma1, f32, "sigma1"),
                          cov0=q(cov0, torch.float64, "cov0"), cov1=q(cov1, torch.float64, "cov1"), pos_Tw=q(pos_Tw, f32, "pos_Tw"),
                          cov0_world=q(cov0_world, torch.float64, "cov0_world"), color=q(color, torch.uint8, "color"),
                          K=Kd.data_ptr(), T_BS=Td.data_ptr(), prior_pose=None if pr is None else pr.data_ptr(),
                          baseline=float(baseline), time_ns=int(time_ns), out_frame_idx=None)
        L.check(self.lib.mv_map_append(C.byref(fr), C.byref(self.stores()), ops._stream()), "mv_map_append")
        self._keep = (Kd, Td, pr)
        idx = self.n_frames
        self.n_frames += 1
        self.rows_upper += n
        return idx

    def push_map_points(self, pos_Tw: torch.Tensor, cov: torch.Tensor, color: torch.Tensor | None = None) -> None:
        """Dense-mapping tail (Odometry/MACVO.py:329-337): append the map points of the NEWEST frame and its frame2map range."""
        from . import ops

        n = pos_Tw.shape[0]
        self.reserve_map_points(n)
        pos = ops._req(pos_Tw, torch.float32, "pos_Tw")
        cv = ops._req(cov.reshape(n, 9), torch.float64, "cov")
        col = None if color is None else ops._req(color, torch.uint8, "color")
        L.check(self.lib.mv_map_append_points(C.byref(self.stores()), n, pos.data_ptr(), cv.data_ptr(), None if col is None else col.data_ptr(),
                                              ops._stream()), "mv_map_append_points")
        self.map_rows_upper += n

    def reserve_map_points(self, n: int) -> None:
        need = self.map_rows_upper + n
        if need >= self.cap["map_points"]:
            cap = _grow_to(need)
            self._regrow(self.map_points, list(self.map_points), cap)
            self.cap["map_points"] = cap

    def map_point_arrays(self) -> dict[str, np.ndarray]:
        """The map-point store (``VisualMap.map_points``; not part of ``VisualMap.serialize``) as numpy arrays."""
        n = int(self.counts.cpu()[5])
        return {k: v[:n].cpu().numpy() for k, v in self.map_points.items()}

    def set_pose(self, frame_idx: int, pose: torch.Tensor) -> None:
        """``write_graph_data`` (Optimizer.py:104-108): the optimised pose replaces the prior the frame was pushed with."""
        self.frames["pose"][frame_idx].copy_(pose.reshape(7).to(torch.float32), non_blocking=True)

    # ------------------------------------------------------------------ outputs
    def sizes(self) -> tuple[int, int, int, int]:
        c = self.counts.cpu().tolist()          # the one host synchronisation, at write-out time
        if c[4]:
            # the append kernel refused a frame (stores too small for the device-side row offsets) or dropped an edge range the
            # reference would have raised on (Graph.py:183-186): the map is incomplete
            raise L.MacvoHipError(f"device map: {int(c[4])} append(s) refused or edge range(s) dropped (MV_ERR_WORKSPACE): "
                                  "reserve() must cover every frame before it is appended")
        return int(c[0]), int(c[1]), int(c[2]), int(c[3])

    def serialize(self) -> dict[str, np.ndarray]:
        """Same keys / shapes / dtypes as ``VisualMap.serialize`` (VisualMap.py:104-116)."""
        nf, nm, npt, _ = self.sizes()
        out = {}
        for prefix, table, n in (("frames/", self.frames, nf), ("points/", self.points, npt), ("match/", self.match, nm)):
            for k, v in table.items():
                out[f"{prefix}/{k}"] = v[:n].cpu().numpy()      # TensorBundle.serialize: f"{prefix}/{k}" (Graph.py:62-66)
        e = self.edges
        out["edge/frame2match/ranges"] = e["frame2match_ranges"][:nf].cpu().numpy()
        out["edge/frame2match/deg"] = e["frame2match_num"][:nf].cpu().numpy()
        out["edge/point2match/edges"] = e["point2match_edges"][:npt].cpu().numpy()
        out["edge/point2match/deg"] = e["point2match_deg"][:npt].cpu().numpy()
        out["edge/match2point/mapping"] = e["match2point"][:nm].cpu().numpy()
        out["edge/match2frame1/mapping"] = e["match2frame1"][:nm].cpu().numpy()
        out["edge/match2frame2/mapping"] = e["match2frame2"][:nm].cpu().numpy()
        out["edge/frame2map/ranges"] = e["frame2map_ranges"][:nf].cpu().numpy()
        out["edge/frame2map/deg"] = e["frame2map_num"][:nf].cpu().numpy()
        return out

    def body_poses(self) -> torch.Tensor:
        """``T_BS @ pose @ T_BS.Inv()`` per frame (Odometry/Interface.py:47-49), fp32, on the device."""
        from . import ops

        nf = self.n_frames
        out = torch.empty((nf, 7), dtype=torch.float32, device=self.dev)
        L.check(self.lib.mv_body_poses(self.frames["pose"].data_ptr(), self.frames["T_BS"].data_ptr(), nf, out.data_ptr(),
                                       ops._stream()), "mv_body_poses")
        return out

    def poses_array(self) -> np.ndarray:
        """Rows of ``poses.npy``: ``[time_ns, x y z qx qy qz qw]`` float64 (np.concatenate of int64 and float32 columns)."""
        t = self.frames["time_ns"][: self.n_frames].cpu().numpy()[:, np.newaxis]
        return np.concatenate([t, self.body_poses().cpu().numpy()], axis=-1)

    def motion_interpolate(self) -> int:
        """``MotionInterpolate.elaborate_map`` (MapProcessor.py:52-76) on the frame poses, in place; returns the number of
        interpolated motions."""
        from . import ops

        nf = self.n_frames
        if nf < 2:
            return 0
        scratch = torch.empty((nf - 1, 7), dtype=torch.float64, device=self.dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=self.dev)
        L.check(self.lib.mv_motion_interpolate(self.frames["pose"].data_ptr(), self.frames["need_interp"].data_ptr(), nf,
                                               scratch.data_ptr(), cnt.data_ptr(), ops._stream()), "mv_motion_interpolate")
        return int(cnt.item())

    def write(self, folder: str) -> None:
        """``poses.npy`` + ``tensor_map.npz`` of ``Odometry/Interface.py:51-52``."""
        os.makedirs(folder, exist_ok=True)
        np.save(os.path.join(folder, "poses.npy"), self.poses_array())
        np.savez_compressed(os.path.join(folder, "tensor_map.npz"), **self.serialize())
