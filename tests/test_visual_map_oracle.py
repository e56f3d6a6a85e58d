"""oracle/visual_map.py against the REAL reference map classes (tests/golden/visual_map.npz, produced by VisualMap /
TensorBundle / the edge tables / MotionInterpolate run unmodified by tests/golden/make_golden.py::gen_visual_map)."""
import os

import numpy as np
import torch

from oracle import visual_map as VM

HERE = os.path.dirname(os.path.abspath(__file__))


def load_golden():
    z = np.load(os.path.join(HERE, "golden", "visual_map.npz"))
    meta = dict(K=torch.from_numpy(z["meta/K"]), T_BS=torch.from_numpy(z["meta/T_BS"]), baseline=float(z["meta/baseline"]))
    frames = [dict(n=0, time_ns=int(z["meta/time0"]))]
    for t in range(1, int(z["meta/n_frames"])):
        fr = {k: torch.from_numpy(z[f"in/{t}/{k}"]) for k in ("valid", "kp0", "kp1", "vals", "sigma0", "sigma1", "cov0", "cov1",
                                                               "pos_Tw", "cov0w", "color", "prior", "opt", "map_pos_Tw", "map_cov",
                                                               "map_color")}
        fr["n"] = fr["kp0"].shape[0]
        fr["time_ns"] = int(z[f"in/{t}/time_ns"])
        frames.append(fr)
    return z, meta, frames


def assert_serialized_equal(got: dict, z) -> None:
    want = {k[4:]: z[k] for k in z.files if k.startswith("ser/")}
    assert set(got) == set(want), sorted(set(got) ^ set(want))
    for k, w in want.items():
        g = np.asarray(got[k])
        assert g.dtype == w.dtype and g.shape == w.shape, (k, g.dtype, w.dtype, g.shape, w.shape)
        assert np.array_equal(g, w, equal_nan=True), k


def test_oracle_map_equals_real_visualmap():
    z, meta, frames = load_golden()
    m = VM.OracleVisualMap()
    for t, fr in enumerate(frames):
        idx = m.push_frame(meta, fr)
        if t:
            if int(fr["valid"].sum()) >= 10:            # mapping only when tracking succeeded (MACVO.py:303-307, 313-337)
                m.push_map_points(idx, fr["map_pos_Tw"], fr["map_cov"], fr["map_color"])
            m.set_pose(idx, fr["opt"])
    ser = m.serialize()
    assert_serialized_equal(ser, z)
    # the map-point store (not part of VisualMap.serialize) and the frame2map edges that index it
    mp = m.map_point_arrays()
    for k in ("pos_Tw", "cov_Tw", "color"):
        assert mp[k].dtype == z[f"mp/{k}"].dtype and np.array_equal(mp[k], z[f"mp/{k}"]), k
    assert (ser["edge/frame2map/deg"] == np.array([0, 1, 1, 1, 0, 0, 1, 1, 1])).all()          # frames 4, 5 lost track: no map points
    last = ser["edge/frame2map/ranges"][-1, 0]
    assert np.array_equal(np.arange(last[0], last[0] + last[1]), z["mp/project_last"])
    assert ser["frames//need_interp"].sum() == 2 and "frames//K" in ser                     # the lost-track frames are in the golden
    # poses.npy rows (Odometry/Interface.py:47-51): int64 time column + float32 body poses, concatenated -> float64
    P = m.poses_array()
    assert P.dtype == z["poses_npy"].dtype and np.array_equal(P[:, 0], z["poses_npy"][:, 0])
    np.testing.assert_allclose(P[:, 1:], z["poses_npy"][:, 1:], rtol=0, atol=1e-6)
    # MotionInterpolate (real class on the PyPose shim)
    pose, idx = VM.motion_interpolate(torch.from_numpy(ser["frames//pose"]), torch.from_numpy(ser["frames//need_interp"]))
    assert idx.tolist() == z["interp/idx"].tolist() and len(idx) == 2
    np.testing.assert_allclose(pose.numpy(), z["interp/pose"], rtol=0, atol=1e-6)
    # frames that were not interpolated and precede the first interpolated motion keep their pose exactly (cumulative product)
    first = int(idx[0]) + 1
    np.testing.assert_allclose(pose.numpy()[:first], ser["frames//pose"][:first], rtol=0, atol=2e-7)


def test_motion_interpolate_is_identity_without_lost_frames():
    """Self-consistency anchor of the un-pinnable PyPose ``cumops`` operand order: with nothing to interpolate,
    pose[0] @ cumprod(pose[i]^-1 pose[i+1]) must rebuild the input trajectory."""
    z, _, _ = load_golden()
    pose = torch.from_numpy(z["ser/frames//pose"])
    out, idx = VM.motion_interpolate(pose, torch.zeros(pose.shape[0], dtype=torch.bool))
    assert idx.numel() == 0
    np.testing.assert_allclose(out.numpy(), pose.numpy(), rtol=0, atol=3e-7)
