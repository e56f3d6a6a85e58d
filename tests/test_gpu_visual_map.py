"""Device-resident VisualMap (mac-vo_amd/devmap.py, csrc/visual_map.hip) against the golden the REAL
reference classes produced, and end to end behind the native frame driver."""
import numpy as np
import pytest
import torch

from tests.test_visual_map_oracle import assert_serialized_equal, load_golden

pytestmark = pytest.mark.gpu


def _push_golden(gpu, init_size):
    from macvo_amd import ops
    from macvo_amd.devmap import DeviceVisualMap

    z, meta, frames = load_golden()
    m = DeviceVisualMap(gpu, init_size=init_size)
    for t, fr in enumerate(frames):
        if t == 0:
            m.push_frame(K=meta["K"], T_BS=meta["T_BS"], baseline=meta["baseline"], time_ns=fr["time_ns"])
            continue
        d = lambda k: fr[k].to(gpu)  # noqa: E731
        tr = ops.TrackedKeypoints(d("kp0"), d("kp1"), None, d("vals").contiguous(), d("sigma0"), d("sigma1"))
        idx = m.push_frame(K=meta["K"], T_BS=meta["T_BS"], baseline=meta["baseline"], time_ns=fr["time_ns"], prior_pose=fr["prior"],
                           tracked=tr, valid=d("valid"), cov0=d("cov0"), cov1=d("cov1"), pos_Tw=d("pos_Tw"),
                           cov0_world=d("cov0w"), color=d("color"))
        if int(fr["valid"].sum()) >= 10:                # dense-mapping tail: tracked frames only (MACVO.py:303-307, 313-337)
            m.push_map_points(d("map_pos_Tw"), d("map_cov"), d("map_color"))
        m.set_pose(idx, fr["opt"].to(gpu))
    return z, m


@pytest.mark.parametrize("init_size", [1024, 16])        # 16: every store re-grows several times (AutoScalingTensor semantics)
def test_device_map_equals_real_visualmap(gpu, init_size, tmp_path):
    z, m = _push_golden(gpu, init_size)
    torch.cuda.synchronize()
    assert_serialized_equal(m.serialize(), z)                       # every store and edge table (frame2map included), bit for bit
    mp = m.map_point_arrays()                                       # VisualMap.map_points: not serialized by the reference, recorded beside it
    for k in ("pos_Tw", "cov_Tw", "color"):
        assert mp[k].dtype == z[f"mp/{k}"].dtype and np.array_equal(mp[k], z[f"mp/{k}"]), k
    nf, nm, npt, lost = m.sizes()
    assert (nf, nm, npt, lost) == (9, z["ser/match//pixel1_uv"].shape[0], z["ser/points//pos_Tw"].shape[0], 2)
    P = m.poses_array()
    assert P.dtype == np.float64 and np.array_equal(P[:, 0], z["poses_npy"][:, 0])
    np.testing.assert_allclose(P[:, 1:], z["poses_npy"][:, 1:], rtol=0, atol=1e-6)
    # the writers of Odometry/Interface.py:51-52
    m.write(str(tmp_path))
    np.testing.assert_array_equal(np.load(tmp_path / "poses.npy"), P)
    back = np.load(tmp_path / "tensor_map.npz")
    assert set(back.files) == {k[4:] for k in z.files if k.startswith("ser/")}
    # MotionInterpolate on the device vs the real class (PyPose shim) and the oracle
    n = m.motion_interpolate()
    assert n == 2
    np.testing.assert_allclose(m.frames["pose"][:nf].cpu().numpy(), z["interp/pose"], rtol=0, atol=1e-6)


def test_motion_interpolate_identity_and_long_track(gpu):
    from macvo_amd.devmap import DeviceVisualMap
    from oracle import se3
    from oracle import visual_map as VM

    g = torch.Generator().manual_seed(3)
    T = 700
    xi = torch.cat([0.1 + 0.02 * torch.randn(T, 3, generator=g), 0.02 * torch.randn(T, 3, generator=g)], -1).double()
    poses = [torch.tensor([0, 0, 0, 0, 0, 0, 1.0], dtype=torch.float64)]
    for k in range(T - 1):
        poses.append(se3.se3_mul(poses[-1], se3.se3_exp(xi[k])))
    pose = torch.stack(poses).float()
    flags = torch.zeros(T, dtype=torch.bool)
    flags[[5, 6, 7, 100, 350, 351, 698]] = True                     # runs of lost frames; 698 sits in the protected tail
    m = DeviceVisualMap(gpu, init_size=1024)
    m.n_frames = T
    m.frames["pose"][:T] = pose.to(gpu)
    m.frames["need_interp"][:T] = flags.to(gpu)
    n = m.motion_interpolate()
    want, idx = VM.motion_interpolate(pose, flags)
    assert n == idx.numel() == 6
    got = m.frames["pose"][:T].cpu()
    dt = max(se3.pose_error(want[i].double(), got[i].double())[0] for i in range(0, T, 7))
    assert dt < 2e-4 * 70, dt                                        # float32 track of ~70 m: both sides round every pose to fp32
    torch.testing.assert_close(got[:5], pose[:5], rtol=0, atol=3e-7)
    m.frames["pose"][:T] = pose.to(gpu)
    m.frames["need_interp"][:T] = False
    assert m.motion_interpolate() == 0
    torch.testing.assert_close(m.frames["pose"][:T].cpu(), pose, rtol=0, atol=2e-5)   # identity up to fp32 accumulation


def test_native_driver_fills_the_device_map(gpu, tmp_path):
    """NativeHotPath.attach_map: the frames of a software-pipelined stream land in the device-resident map straight from the
    tracking kernels' tables.  Expected content = the same stream's per-frame results pushed through the CPU oracle map."""
    from macvo_amd.devmap import DeviceVisualMap
    from macvo_amd.pipeline import Camera, FrameInputs, HotPathConfig, NativeHotPath
    from oracle import visual_map as VM
    from tests import synth

    n_frames = 12
    cam, frames, _ = synth.make_sequence(n_frames, 240, 320, C=32, iters=2, seed=8)
    for fr in frames[6:8]:                          # two frames with unusable flow covariance -> no candidates -> lost track
        fr["logcov"] = fr["logcov"].clone()
        fr["logcov"][1] = 4.0
    K = torch.tensor([[cam["fx"], 0, cam["cx"]], [0, cam["fy"], cam["cy"]], [0, 0, 1.0]])
    T_BS = torch.tensor([0.05, 0.0, -0.1, 0.0, 0.0, 0.0, 1.0])
    ins = [FrameInputs(**{k: v.to(gpu) for k, v in fr.items()}, time_ns=1000 + 33 * t) for t, fr in enumerate(frames)]
    torch.cuda.synchronize()
    hot = NativeHotPath(Camera(**cam), HotPathConfig(), gpu, keep_extras=True)
    dmap = DeviceVisualMap(gpu, init_size=64)       # small: the stores grow while the pipe is running
    hot.attach_map(dmap, K, T_BS)
    torch.manual_seed(12)
    hot.initialize(ins[0])
    meta = dict(K=K, T_BS=T_BS, baseline=cam["baseline"])
    ora = VM.OracleVisualMap()
    ora.push_frame(meta, dict(n=0, time_ns=1000))
    prior = torch.tensor([0, 0, 0, 0, 0, 0, 1.0])
    sink = torch.zeros(n_frames - 1, 7, device=gpu)
    for t, r in enumerate(hot.run(ins[1:], pose_sink=sink), start=1):
        hot.sync_pose()
        if r.n_sel:
            ex, tr = r.extras, r.extras["tracked"]
            fr = dict(n=r.n_sel, valid=ex["valid"].cpu().clone(), kp0=tr.kp0_uv.cpu().clone(), kp1=tr.kp1_uv.cpu().clone(),
                      vals=tr.vals.cpu().clone(), sigma0=tr.sigma0.cpu().clone(), sigma1=tr.sigma1.cpu().clone(),
                      cov0=ex["cov0"].cpu().clone(), cov1=ex["cov1"].cpu().clone(), pos_Tw=ex["pos_Tw"].cpu().clone(),
                      cov0w=ex["cov0_w"].cpu().clone(), color=None)
        else:
            fr = dict(n=0)
        fr.update(time_ns=1000 + 33 * t, prior=prior.clone())
        idx = ora.push_frame(meta, fr)
        pose = r.pose.cpu().clone()
        ora.set_pose(idx, pose)
        prior = pose
    torch.cuda.synchronize()
    want = ora.serialize()
    got = dmap.serialize()
    assert set(got) == set(want)
    for k, w in want.items():
        assert got[k].dtype == w.dtype and np.array_equal(got[k], w, equal_nan=True), k
    assert got["frames//need_interp"].tolist() == [False] * 6 + [True, True] + [False] * 4
    assert np.array_equal(got["frames//pose"][1:], sink.cpu().numpy())          # optimised poses written over the priors
    np.testing.assert_allclose(dmap.poses_array()[:, 1:], ora.poses_array()[:, 1:], rtol=0, atol=1e-6)
    dmap.write(str(tmp_path))
    assert np.load(tmp_path / "poses.npy").shape == (n_frames, 8)


def test_native_driver_dense_mapping_tail(gpu):
    """`mapping: true` (Config/Experiment/MACVO/MACVO_Fast.yaml:61; Odometry/MACVO.py:313-337) in the NATIVE frame driver: the
    MappingPointSelector on the previous frame's depth maps, the second randperm of the frame (only when tracking succeeded), the
    map-point tables and — with a device map attached — map_points.push + frame2map.add.  Bit for bit against the Python-sequenced
    HotPath (itself checked against the oracle in test_gpu_pipeline) frame by frame, and the device map against the oracle map
    fed with the same per-frame results."""
    from macvo_amd.devmap import DeviceVisualMap
    from macvo_amd.pipeline import Camera, FrameInputs, HotPath, HotPathConfig, NativeHotPath
    from oracle import visual_map as VM
    from tests import synth

    n_frames, H, W = 8, 240, 320
    cam, frames, _ = synth.make_sequence(n_frames, H, W, C=32, iters=2, seed=18)
    g = torch.Generator().manual_seed(4)
    for t, fr in enumerate(frames):
        fr["image"] = torch.rand(1, 3, H, W, generator=g)
        fr["flow"] = fr["flow"].clone()
    frames[5]["logcov"] = frames[5]["logcov"].clone()
    frames[5]["logcov"][1] = 4.0                       # frame 5: no tracking candidates -> lost track -> no mapping, no second randperm
    cfg = dict(mapping=True, map_num_point=300, map_max_depth=40.0, map_max_depth_cov=50.0, num_point=120)
    K = torch.tensor([[cam["fx"], 0, cam["cx"]], [0, cam["fy"], cam["cy"]], [0, 0, 1.0]])
    T_BS = torch.tensor([0, 0, 0, 0, 0, 0, 1.0])
    ins = [FrameInputs(**{k: v.to(gpu) for k, v in fr.items()}, time_ns=50 + t) for t, fr in enumerate(frames)]
    torch.cuda.synchronize()
    py = HotPath(Camera(**cam), HotPathConfig(**cfg), gpu)
    nat = NativeHotPath(Camera(**cam), HotPathConfig(**cfg), gpu, keep_extras=True)
    dmap = DeviceVisualMap(gpu, init_size=256)
    nat.attach_map(dmap, K, T_BS)
    py.initialize(ins[0])
    nat.initialize(ins[0])
    meta = dict(K=K, T_BS=T_BS, baseline=cam["baseline"])
    ora = VM.OracleVisualMap()
    ora.push_frame(meta, dict(n=0, time_ns=50))
    prior = torch.tensor([0, 0, 0, 0, 0, 0, 1.0])
    n_mapped = 0
    for t in range(1, n_frames):
        torch.manual_seed(600 + t)
        a = py.step(ins[t])
        st_a = torch.get_rng_state()
        torch.manual_seed(600 + t)
        b = nat.step(ins[t])
        torch.cuda.synchronize()
        assert torch.equal(torch.get_rng_state(), st_a), t                 # both consumed the global CPU generator identically
        assert torch.equal(a.kp0_uv, b.kp0_uv) and torch.equal(a.pose, b.pose), t
        assert (a.map_points is None) == (b.map_points is None), t
        if a.map_points is not None:
            n_mapped += 1
            for f in ("uv", "depth", "sigma_dd", "pos_Tc", "pos_Tw", "cov_Tc", "color"):
                assert torch.equal(getattr(a.map_points, f), getattr(b.map_points, f)), (t, f)
            assert 0 < b.map_points.uv.shape[0] <= 300
        if b.n_sel:
            ex, tr = b.extras, b.extras["tracked"]
            fr = dict(n=b.n_sel, valid=ex["valid"].cpu().clone(), kp0=tr.kp0_uv.cpu().clone(), kp1=tr.kp1_uv.cpu().clone(),
                      vals=tr.vals.cpu().clone(), sigma0=tr.sigma0.cpu().clone(), sigma1=tr.sigma1.cpu().clone(),
                      cov0=ex["cov0"].cpu().clone(), cov1=ex["cov1"].cpu().clone(), pos_Tw=ex["pos_Tw"].cpu().clone(),
                      cov0w=ex["cov0_w"].cpu().clone(), color=None)
        else:
            fr = dict(n=0)
        fr.update(time_ns=50 + t, prior=prior.clone())
        idx = ora.push_frame(meta, fr)
        if b.map_points is not None:
            mpb = b.map_points
            ora.push_map_points(idx, mpb.pos_Tw.cpu().clone(), mpb.cov_Tc.cpu().clone(), mpb.color.cpu().clone())
        pose = b.pose.cpu().clone()
        ora.set_pose(idx, pose)
        prior = pose
        py.pose = b.pose.clone()
    assert n_mapped == n_frames - 2                                         # every frame but the lost one
    want, got = ora.serialize(), dmap.serialize()
    for k, w in want.items():
        assert got[k].dtype == w.dtype and np.array_equal(got[k], w, equal_nan=True), k
    assert got["edge/frame2map/deg"].tolist() == [0, 1, 1, 1, 1, 0, 1, 1]
    mp_w, mp_g = ora.map_point_arrays(), dmap.map_point_arrays()
    for k in mp_w:
        assert np.array_equal(mp_w[k], mp_g[k]), k
