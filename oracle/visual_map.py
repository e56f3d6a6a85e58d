"""CPU restatement of the tracking-map registration, its outputs and MotionInterpolate (TEST INFRASTRUCTURE).

Follows, with plain numpy / torch:
* ``Odometry/MACVO.py:158-171`` (initialize: push the first frame) and ``:244-311`` (run_pair: ``MatchObs.init`` -> ``[mask]``,
  ``points.push(PointNode.init(...)[mask])``, ``push_keyframe`` :339-347, the six edge updates :286-293, the lost-track flag
  :303-307), on the stores of ``Module/Map/VisualMap.py:15-102`` and the edge tables of ``Module/Map/Graph.py:133-298``;
* ``VisualMap.serialize`` (:104-116) — same keys (the reference's ``frames//K`` double slash included);
* ``Odometry/Interface.py:47-51`` — rows of ``poses.npy``;
* ``Module/MapProcessor.py:52-76`` (``MotionInterpolate.elaborate_map``) with ``Utility/Math.py:96-133``.
Pinned by tests/golden/visual_map.npz, produced by the REAL ``VisualMap`` / ``MotionInterpolate`` classes
(tests/golden/make_golden.py::gen_visual_map; the PyPose calls of the latter run on tests/golden/pypose_shim.py, so the
SE3 Log / Exp / cumops pieces stay "parity unpinned").
"""
from __future__ import annotations

import numpy as np
import torch

from . import se3

MATCH_FROM_VALS = {"pixel1_d": 0, "pixel1_disp": 1, "pixel1_disp_cov": 2, "pixel1_d_cov": 3, "pixel2_d": 4, "pixel2_disp": 5,
                   "pixel2_disp_cov": 6, "pixel2_d_cov": 7}


class OracleVisualMap:
    def __init__(self, max_pt_obs: int = 5, max_frame_range: int = 2):
        self.max_pt_obs, self.max_frame_range = max_pt_obs, max_frame_range
        self.frames = {k: [] for k in ("K", "baseline", "pose", "T_BS", "need_interp", "time_ns")}
        self.points = {k: [] for k in ("pos_Tw", "cov_Tw", "color")}
        self.match = {k: [] for k in ("pixel1_uv", "pixel2_uv", "pixel1_d", "pixel2_d", "pixel1_disp", "pixel2_disp",
                                      "pixel1_disp_cov", "pixel2_disp_cov", "obs1_covTc", "obs2_covTc", "pixel1_uv_cov",
                                      "pixel2_uv_cov", "pixel1_d_cov", "pixel2_d_cov")}
        self.f2m_ranges, self.f2m_num, self.f2map_ranges, self.f2map_num = [], [], [], []
        self.m2f1, self.m2f2, self.m2p, self.p2m_edges, self.p2m_deg = [], [], [], [], []
        self.n_match = self.n_points = 0
        self.map_points = {k: [] for k in ("pos_Tw", "cov_Tw", "color")}
        self.n_map_points = 0

    def push_frame(self, meta, fr, min_num_point: int = 10) -> int:
        F = len(self.frames["pose"])
        prev = F - 1
        n = fr["n"]
        mask = fr["valid"] if n else torch.zeros(0, dtype=torch.bool)
        kept = int(mask.sum()) if n else 0
        M0, P0 = self.n_match, self.n_points
        if n:
            rows = {"pixel1_uv": fr["kp0"], "pixel2_uv": fr["kp1"], "pixel1_uv_cov": fr["sigma0"], "pixel2_uv_cov": fr["sigma1"],
                    "obs1_covTc": fr["cov0"], "obs2_covTc": fr["cov1"]}
            rows.update({k: fr["vals"][i].unsqueeze(-1) for k, i in MATCH_FROM_VALS.items()})
            for k in self.match:
                self.match[k].append(rows[k][mask])                                   # match_obs[mask]   (:270)
            self.points["pos_Tw"].append(fr["pos_Tw"][mask])                        # PointNode.init(...)[mask] (:277-281)
            self.points["cov_Tw"].append(fr["cov0w"][mask])
            self.points["color"].append(fr["color"][mask] if fr.get("color") is not None else torch.zeros(kept, 3, dtype=torch.uint8))
        # push_keyframe (:339-347): the frame enters at its prior
        self.frames["K"].append(meta["K"].reshape(1, 3, 3).float())
        self.frames["baseline"].append(torch.tensor([meta["baseline"]], dtype=torch.float32))
        self.frames["pose"].append(fr.get("prior", torch.tensor([0, 0, 0, 0, 0, 0, 1.0])).reshape(1, 7).float())
        self.frames["T_BS"].append(meta["T_BS"].reshape(1, 7).float())
        self.frames["need_interp"].append(torch.tensor([prev >= 0 and kept < min_num_point]))   # :303-307
        self.frames["time_ns"].append(torch.tensor([fr["time_ns"]], dtype=torch.long))
        self.f2m_ranges.append(torch.full((self.max_frame_range, 2), -1, dtype=torch.long))
        self.f2m_num.append(0)
        self.f2map_ranges.append(torch.full((self.max_frame_range, 2), -1, dtype=torch.long))
        self.f2map_num.append(0)
        if prev >= 0:
            for f in (prev, F):                                                       # frame2match.add (:290-291)
                self.f2m_ranges[f][self.f2m_num[f]] = torch.tensor([M0, kept])
                self.f2m_num[f] += 1
            for k in range(kept):
                e = torch.full((self.max_pt_obs,), -1, dtype=torch.long)
                e[0] = M0 + k                                                         # point2match.add (:288)
                self.p2m_edges.append(e)
                self.p2m_deg.append(1)
                self.m2p.append(P0 + k)                                               # match2point.set (:289)
                self.m2f1.append(prev)                                                # :292
                self.m2f2.append(F)                                                   # :293
        self.n_match += kept
        self.n_points += kept
        return F

    def push_map_points(self, frame_idx: int, pos_Tw: torch.Tensor, cov_Tw: torch.Tensor, color: torch.Tensor | None = None) -> None:
        """Dense-mapping tail of run_pair (Odometry/MACVO.py:329-337): ``map_points.push(...)`` + ``frame2map.add(frame_idx, start, n)``
        (only for frames that kept >= min_num_point observations, :303-307; the covariance is stored unrotated, :324,334)."""
        n = pos_Tw.shape[0]
        self.map_points["pos_Tw"].append(pos_Tw.float())
        self.map_points["cov_Tw"].append(cov_Tw.double())
        self.map_points["color"].append(color if color is not None else torch.zeros(n, 3, dtype=torch.uint8))
        k = self.f2map_num[frame_idx]
        assert k < self.max_frame_range, "DenseEdge_Multi.add: no free range slot"         # the reference raises here (Graph.py:183-186)
        self.f2map_ranges[frame_idx][k] = torch.tensor([self.n_map_points, n])
        self.f2map_num[frame_idx] = k + 1
        self.n_map_points += n

    def map_point_arrays(self) -> dict:
        shapes = {"pos_Tw": ((0, 3), torch.float32), "cov_Tw": ((0, 3, 3), torch.float64), "color": ((0, 3), torch.uint8)}
        return {k: (torch.cat(v) if v else torch.zeros(shapes[k][0], dtype=shapes[k][1])).numpy() for k, v in self.map_points.items()}

    def set_pose(self, idx: int, pose: torch.Tensor) -> None:                         # write_graph_data (Optimizer.py:104-108)
        self.frames["pose"][idx] = pose.reshape(1, 7).float()

    def serialize(self) -> dict:
        cat = lambda lst, empty: torch.cat(lst) if lst else empty  # noqa: E731
        out = {}
        for k, v in self.frames.items():
            out[f"frames//{k}"] = torch.cat(v).numpy()
        shapes = {"pos_Tw": ((0, 3), torch.float32), "cov_Tw": ((0, 3, 3), torch.float64), "color": ((0, 3), torch.uint8)}
        for k, v in self.points.items():
            out[f"points//{k}"] = cat(v, torch.zeros(shapes[k][0], dtype=shapes[k][1])).numpy()
        for k, v in self.match.items():
            out[f"match//{k}"] = torch.cat(v).numpy()
        i64 = lambda x: np.asarray(x, dtype=np.int64)  # noqa: E731
        out["edge/frame2match/ranges"] = torch.stack(self.f2m_ranges).numpy()
        out["edge/frame2match/deg"] = i64(self.f2m_num)
        out["edge/point2match/edges"] = torch.stack(self.p2m_edges).numpy() if self.p2m_edges else np.zeros((0, self.max_pt_obs), np.int64)
        out["edge/point2match/deg"] = i64(self.p2m_deg)
        out["edge/match2point/mapping"] = i64(self.m2p)
        out["edge/match2frame1/mapping"] = i64(self.m2f1)
        out["edge/match2frame2/mapping"] = i64(self.m2f2)
        out["edge/frame2map/ranges"] = torch.stack(self.f2map_ranges).numpy()
        out["edge/frame2map/deg"] = i64(self.f2map_num)
        return out

    def poses_array(self) -> np.ndarray:
        """Interface.py:47-51: ``[time_ns, T_BS @ pose @ T_BS.Inv()]`` (float32 SE3 arithmetic, float64 rows)."""
        pose, tbs = torch.cat(self.frames["pose"]), torch.cat(self.frames["T_BS"])
        body = se3.se3_mul(se3.se3_mul(tbs, pose), se3.se3_inv(tbs))
        t = torch.cat(self.frames["time_ns"]).numpy()[:, np.newaxis]
        return np.concatenate([t, body.numpy()], axis=-1)


def motion_interpolate(pose: torch.Tensor, need_interp: torch.Tensor):
    """``MotionInterpolate.elaborate_map`` (MapProcessor.py:57-76) -> (new poses [T,7] float32, interpolated motion indices)."""
    P = pose.double()
    bad = need_interp[1:].bool().clone()
    motions = se3.se3_mul(se3.se3_inv(P[:-1]), P[1:])
    bad[:2] = False
    bad[-2:] = False
    idx = torch.nonzero(bad).flatten()
    good = torch.nonzero(~bad).flatten()
    for i in idx.tolist():                                                            # interpolate_pose (Math.py:96-121)
        e = int(torch.searchsorted(good, torch.tensor(i), right=False))
        s_i, e_i = int(good[e - 1]), int(good[e])
        prop = (torch.tensor(i - s_i) / torch.tensor(e_i - s_i)).double()             # int64 / int64 -> float32 (:114)
        diff = se3.se3_log(se3.se3_mul(motions[e_i], se3.se3_inv(motions[s_i])))
        motions[i] = se3.se3_mul(se3.se3_exp(prop * diff), motions[s_i])
    norm = lambda x: torch.cat([x[:3], x[3:] / x[3:].norm()])  # noqa: E731  NormalizeQuat (Math.py:124-133)
    out = [P[0]]
    run = motions[0]
    for k in range(motions.shape[0]):                                                 # cumops with NormalizeQuat on both operands (:73)
        run = motions[0] if k == 0 else se3.se3_mul(norm(run), norm(motions[k]))
        out.append(se3.se3_mul(P[0], run))
    return torch.stack(out).float(), idx
