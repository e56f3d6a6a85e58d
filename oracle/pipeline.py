"""CPU restatement of one ``MACVO.run_pair`` over the hot path (TEST INFRASTRUCTURE).

Chains the oracle pieces in the order of ``Odometry/MACVO.py:173-311`` so that tests can compare the HIP
pipeline frame by frame and ``bench.py`` can time a CPU baseline shaped like the reference (torch einsum volume,
``grid_sample`` lookup with a materialised grid, ``max_pool2d`` + ``nonzero`` selector, gather + einsum covariance,
dense-``block_diag`` float64 LM).  Inputs are the same :class:`FrameInputs`-like tensors, on the CPU.
"""
from __future__ import annotations

import torch

from . import corr, covariance, filters, frontend, pgo, se3, selector


def rotation_matrix_f32(pose: torch.Tensor) -> torch.Tensor:
    """``pp.SE3(pose).rotation().matrix()`` in the pose dtype: PyPose builds it as ``Act(I).T`` (columns = q.Act(e_i))."""
    q = pose[3:]
    cols = se3.quat_act(q, torch.eye(3, dtype=pose.dtype))
    return cols.transpose(-1, -2).contiguous()


class OracleHotPath:
    def __init__(self, cam: dict, cfg: dict | None = None):
        self.cam = cam
        d = dict(num_point=200, edgewidth=32, match_cov_default=0.25, selector="nodepth", kp_kernel_size=7,
                 kp_mask_width=32, max_match_cov=100.0, max_depth_cov=250.0, max_depth="auto", cov_kernel_size=31,
                 min_flow_cov=0.25, min_depth_cov=0.05, graph_type="disp", min_num_point=10, radius=4,
                 mapping=False, map_num_point=2000, map_max_depth=5.0, map_max_depth_cov=0.005, map_mask_width=32,
                 volume_store="fp32")   # "encoder": Fast mode, the einsum's fp16 result widened (flownet.py:26-27 with enc_dtype fp16)
        d.update(cfg or {})
        self.cfg = d
        self.maps_prev = None
        self.pose = torch.tensor([0, 0, 0, 0, 0, 0, 1], dtype=torch.float32)
        self.last_tokens = None
        self._prev_image = None
        self.timing: dict = {}

    def frontend(self, x: dict) -> dict:
        f1, f2 = x["fmap1"].float(), x["fmap2"].float()
        vol = corr.corr_volume(f1, f2, torch.float32)
        if self.cfg["volume_store"] == "encoder":
            vol = vol.to(torch.float16).float()
        for it in range(x["coords"].shape[0]):
            self.last_tokens = corr.corr_lookup(vol, x["coords"][it], self.cfg["radius"])
        if x.get("flow8") is not None:                                                 # covhead.py:119-135
            flow = frontend.upsample_flow(x["flow8"], 0.25 * x["up_mask"])
            cov = torch.exp(frontend.upsample_flow(x["cov8"], x["cov_mask"]) * 2)
        elif x.get("cov_exp") is not None:                                             # tests: covariance already exponentiated
            flow, cov = x["flow"], x["cov_exp"]
        else:
            flow = x["flow"]
            cov = torch.exp(x["logcov"] * 2)                                           # flownet.py:44
        depth, depth_cov, disp, disp_cov, _ = frontend.inference_2_depth(flow[0:1], cov[0:1], self.cam["baseline"], self.cam["fx"])
        return dict(depth=depth, cov=depth_cov, disparity=disp, disparity_uncertainty=disp_cov,
                    flow=flow[1:2], flow_cov=frontend.from_partial_cov(cov[1:2]))

    def initialize(self, x: dict, init_pose=None):
        self.maps_prev = self.frontend(x)
        self._prev_image = x.get("image")
        if init_pose is not None:
            self.pose = init_pose.float().reshape(7).clone()

    def step(self, x: dict) -> dict:
        c, cam = self.cfg, self.cam
        H, W = cam["H"], cam["W"]
        K4 = (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
        Km = torch.tensor([[cam["fx"], 0, cam["cx"]], [0, cam["fy"], cam["cy"]], [0, 0, 1]], dtype=torch.float32)
        maps0, maps1 = self.maps_prev, self.frontend(x)
        if c["selector"] == "nodepth":
            kp0, cand, _ = selector.cov_aware_selector_nodepth(maps1["flow_cov"], c["num_point"], c["kp_kernel_size"],
                                                               c["kp_mask_width"], c["max_match_cov"])
        else:
            md = cam["fx"] * cam["baseline"] if c["max_depth"] == "auto" else c["max_depth"]
            kp0, cand, _ = selector.cov_aware_selector(maps0["depth"], maps0["cov"], maps1["depth"], maps1["cov"],
                                                       maps1["flow_cov"], c["num_point"], md, c["kp_kernel_size"],
                                                       c["kp_mask_width"], c["max_depth_cov"], c["max_match_cov"])
        kp0_all = kp0
        tr = frontend.track_keypoints(kp0, maps1["flow"], maps1["flow_cov"], maps0, maps1, c["edgewidth"], H, W,
                                      c["match_cov_default"])
        kp0, kp1 = tr["kp0_uv"], tr["kp1_uv"]
        pos0_Tc = frontend.pixel2point_NED(kp0, tr["kp0_d"], Km)
        mc = dict(kernel_size=c["cov_kernel_size"], match_cov_default=c["match_cov_default"],
                  min_flow_cov=c["min_flow_cov"], min_depth_cov=c["min_depth_cov"])
        cov0 = covariance.match_covariance(kp0, maps0["depth"], tr["kp0_sigma_dd"], tr["kp0_sigma_uv"], *K4, **mc)
        cov1 = covariance.match_covariance(kp1, maps1["depth"], tr["kp1_sigma_dd"], tr["kp1_sigma_uv"], *K4, **mc)
        # CovarianceSanityFilter (OutlierFilter.py:91-100)
        mask = filters.covariance_sanity(cov0, cov1)
        R = rotation_matrix_f32(self.pose)
        pos_Tw = se3.se3_act(self.pose, pos0_Tc)                                      # fp32 (MACVO.py:277)
        cov_Tw = covariance.rotate_covariance(R, cov0)
        n = int(mask.sum())
        out = dict(kp0_uv=kp0_all, n_valid=n)
        prev_pose = self.pose.clone()
        if n >= c["min_num_point"]:
            prob = pgo.PGOProblem(init_pose=self.pose.clone(), K=Km, baseline=cam["baseline"], pos_Tw=pos_Tw[mask],
                                  cov_Tw=cov_Tw[mask], pixel2_uv=kp1[mask], pixel2_d=tr["kp1_d"][mask].unsqueeze(-1),
                                  pixel2_disp=tr["kp1_disparity"].T[mask], pixel2_disp_cov=tr["kp1_sigma_disparity"].T[mask],
                                  pixel2_uv_cov=tr["kp1_sigma_uv"][mask], obs2_covTc=cov1[mask])
            res = pgo.solve(prob, c["graph_type"])
            self.pose = res.pose.float()                                               # write_graph_data
            out.update(pose_f64=res.pose, steps=res.steps, loss=res.loss)
            if c["mapping"]:                                                           # MACVO.py:313-337 (skipped when lost, :303-307)
                muv, _, _ = selector.mapping_point_selector(maps0["depth"], maps0["cov"], c["map_num_point"], c["map_max_depth"],
                                                            c["map_max_depth_cov"], c["map_mask_width"])
                md = frontend.retrieve_pixels(muv, maps0["depth"]).squeeze(0)
                m_Tc = frontend.pixel2point_NED(muv, md, Km)
                m_sdd = frontend.retrieve_pixels(muv, maps0["cov"]).squeeze(0)
                m_suv = torch.ones((muv.shape[0], 3)) * c["match_cov_default"]
                m_suv[..., 2] = 0.
                m_cov = covariance.match_covariance(muv, maps0["depth"], m_sdd, m_suv, *K4, **mc)   # stored unrotated (:324,334)
                out["map"] = dict(uv=muv, depth=md, sigma_dd=m_sdd, pos_Tc=m_Tc, pos_Tw=se3.se3_act(prev_pose, m_Tc), cov_Tc=m_cov)
                if self._prev_image is not None:                                   # frame0.stereo.imageL (:327)
                    img = self._prev_image.reshape(1, 3, H, W)
                    out["map"]["color"] = (img[..., muv[..., 1], muv[..., 0]].squeeze(0).T * 255).to(torch.uint8)
        out["pose"] = self.pose
        out.update(cov0=cov0, cov1=cov1, cov_Tw=cov_Tw, pos_Tw=pos_Tw, mask=mask, tracked=tr)
        self.maps_prev = maps1
        self._prev_image = x.get("image")
        return out
