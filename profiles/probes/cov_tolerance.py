"""Scratch: measured relative error of mv_match_cov vs the reference goldens / the oracle (to size the test tolerances)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from macvo_amd import ops
from tests.test_gpu_golden import load
gpu = torch.device("cuda:0")
z = load("covariance"); K = [float(v) for v in z["K"]]; depth = z["depth"].to(gpu)
def rel(a, b):
    d = (a - b).abs(); return float((d / b.abs().clamp_min(1e-30)).max()), float(d.max())
out = ops.match_cov(depth, z["kp_int"].to(gpu), z["flow_cov_in"].clone().to(gpu), None, *K); print("int_flowcov", rel(out.cpu(), z["cov_int_flowcov"]))
out = ops.match_cov(depth, z["kp_float"].to(gpu), z["flow_cov_in"].clone().to(gpu), None, *K); print("float_flowcov", rel(out.cpu(), z["cov_float_flowcov"]))
s0 = torch.ones(z["kp_int"].shape[0], 3) * 0.25; s0[:, 2] = 0
out = ops.match_cov(depth, z["kp_int"].to(gpu), s0.to(gpu), z["depth_cov_kp"].to(gpu), *K, use_patch_var=True); print("default_sigma", rel(out.cpu(), z["cov_int_default_sigma"]))
out = ops.match_cov(depth, z["kp_int"].to(gpu), s0.to(gpu), z["depth_cov_kp"].to(gpu), *K, use_patch_var=False); print("nodefault", rel(out.cpu(), z["cov_int_nodefault"]))
