#!/usr/bin/env python
"""End to end on the MI355X: stereo IMAGES -> poses through the reference's own, unmodified ``Odometry/MACVO.py`` loop (``from_config`` ->
``receive_frames`` -> ``initialize`` / ``run_pair``, driven by tests/refrun.py on ``oracle/_ref/pyref``), with

  * the learned frontend = a FlowFormerCov-SHAPED network in PyTorch-ROCm (tools/flowformer_host.py: the published FlowFormer
    architecture with the in-tree ``covhead.py`` / ``flownet.py`` statement order; RANDOM weights — the submodule's source and the
    released checkpoint are absent from the reference checkout), behind ``HIP_FlowFormerCovFrontend(config.model = ...)``;
  * ``--variant hooked``: ``plugins.install_flowformer_hooks`` routes the all-pairs volume, the 12 window lookups, the 24 convex upsamplings
    and the cost patch embedding of that network through the HIP library; ``--variant unhooked``: the same network as plain PyTorch ops
    (einsum / grid_sample / softmax-unfold / MIOpen convolutions) — the "before";
  * selector, covariance model and pose-graph solve = the ``HIP_*`` plugins; map, motion model, keyframes, filters, writers = the reference's code.

What the number is and is not.  The dtype switches are ``MACVO_Fast.yaml:69-76`` (encoder fp16, decoder bf16, 12 decoder iterations) — the
configuration the reference quotes 12.5 frames/s for on an RTX 6000 Ada (README.md:28,117).  Weights are random, so the network's output is
meaningless as a flow field; to give the backend its real work (200 tracked keypoints, covariances, a converging solve) the network's
(flow, sigma) output is ADDED to / MULTIPLIED onto the synthetic scene's fields of tests/synth.make_sequence (the benchmark's stream), frame
index in pixel (0, 0) as in tests/refrun.py's ReplayNet: every layer of the network and every kernel of the hot path runs, on data-dependent
inputs, once per frame; accuracy is not measured here.  Measurement plumbing — not part of the drop-in.

    python tools/end_to_end.py --frames 24 --warmup 4 --variants hooked,unhooked          # prints ONE JSON line
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402

DT = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}


class PriorNet:
    """``inference(A, B)`` of the host network on top of the synthetic scene's fields (see the module docstring).  Exposes nothing but
    ``inference`` / ``eval``, so the plugin does not install hooks a second time (the caller decides: hooked or not)."""

    def __init__(self, host, maps, device):
        self.host = host
        self.flow = torch.stack([f["flow"] for f in maps]).to(device)      # [T, 2, 2, H, W]
        self.cov = torch.stack([f["cov"] for f in maps]).to(device)

    def inference(self, A, B):
        # the frame index is decoded ON THE DEVICE (no host round trip): the same code is valid inside the hipGraph that
        # HIP_CUDAGraph_FlowFormerCovFrontend captures once and replays on its static input buffers
        t = (A[0, 0, 0, 0].float() * 255.0).round().long().clamp(0, self.flow.shape[0] - 1).view(1)
        flow, cov = self.host.inference(A, B)
        n = A.shape[0]
        return self.flow.index_select(0, t)[0, :n] + flow.float(), self.cov.index_select(0, t)[0, :n] * cov.float()

    def eval(self):
        return self


def network_ms(host, H, W, iters=8):
    """one joint stereo + temporal inference (B = 2 pairs, as FlowFormerCovFrontend.estimate_pair batches them, Frontend.py:219-224)"""
    dev = next(host.parameters()).device
    g = torch.Generator().manual_seed(5)
    a, b = torch.rand(2, 3, H, W, generator=g).to(dev), torch.rand(2, 3, H, W, generator=g).to(dev)
    with torch.inference_mode():
        for _ in range(2):
            host.inference(a, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            host.inference(a, b)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def run_variant(ref, refrun, variant: str, args) -> dict:
    import flowformer_host as fh
    from macvo_amd import plugins

    dev = torch.device("cuda:0")
    torch.manual_seed(7)
    host = fh.FlowFormerCovHost(fh.demo_cfg(decoder_depth=args.decoder_depth), DT[args.enc], DT[args.dec]).to(dev).eval()
    if args.channels_last:      # NHWC weights (and, through them, activations): MIOpen's bf16 / fp16 convolutions run in NHWC and otherwise transpose around every call
        host = host.to(memory_format=torch.channels_last)
    hooks = plugins.install_flowformer_hooks(host) if variant == "hooked" else []
    out = {"hooks": hooks, "network_ms_per_pair_batch": round(network_ms(host, args.height, args.width), 3)}

    cam, maps, poses = refrun.synthetic_maps(args.frames)
    assert (cam["H"], cam["W"]) == (args.height, args.width), "the synthetic stream is 640x480"
    frames = refrun.make_stereo_frames(ref, cam, maps, poses)
    case = dict(refrun.CASES["synth_fast"])
    case["mapping"] = bool(args.mapping)
    cfg = refrun.make_config(case, "hip")
    fe = cfg.Odometry.frontend
    fe.type = "HIP_CUDAGraph_FlowFormerCovFrontend" if args.graph else "HIP_FlowFormerCovFrontend"
    fe.args.enc_dtype, fe.args.dec_dtype, fe.args.decoder_depth = args.enc, args.dec, args.decoder_depth
    ref.OM.MACVO.is_valid_config(cfg.Odometry)
    fe.args.model = PriorNet(host, maps, dev)
    from Utility.PrettyPrint import GlobalConsole
    GlobalConsole.quiet = True
    torch.manual_seed(1234)
    system = ref.OM.MACVO.from_config(cfg)
    stamps = [time.perf_counter()]

    def on_frame(frame, sysm, pb):
        torch.cuda.synchronize()
        stamps.append(time.perf_counter())

    with tempfile.TemporaryDirectory() as tmp:
        box = ref.Sandbox(Path(tmp))
        system.receive_frames(frames, box, on_frame_finished=on_frame)
        assert system.terminated and os.path.exists(box.path("poses.npy")), "receive_frames swallowed an exception (see the log above)"
        import numpy as np
        est = np.load(box.path("poses.npy"))
        tm = np.load(box.path("tensor_map.npz"))
        n_match = int(tm["match//pixel1_uv"].shape[0]) if "match//pixel1_uv" in tm.files else -1
    dts = [b - a for a, b in zip(stamps[:-1], stamps[1:])]
    steady = dts[args.warmup:]
    med = sorted(steady)[len(steady) // 2]
    out.update({
        "frames_timed": len(steady), "ms_per_frame": round(1e3 * sum(steady) / len(steady), 3), "fps": round(len(steady) / sum(steady), 2),
        "ms_per_frame_median": round(1e3 * med, 3), "fps_median": round(1.0 / med, 2),     # (single frames of 100+ ms occur: first-use kernel selection in MIOpen / hipBLASLt)
        "ms_per_frame_min": round(1e3 * min(steady), 3), "ms_per_frame_max": round(1e3 * max(steady), 3),
        "tracked_observations": n_match, "poses_written": int(est.shape[0]),
        "classes": {k: type(getattr(system, k)).__name__ for k in ("Frontend", "KeypointSelector", "ObsCovModel", "Optimizer")},
    })
    del system, host
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=4, help="leading frames left out of the average (first frame = MACVO.initialize)")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--enc", choices=list(DT), default="fp16")
    ap.add_argument("--dec", choices=list(DT), default="bf16")
    ap.add_argument("--decoder-depth", type=int, default=12)
    ap.add_argument("--mapping", type=int, default=1, help="dense-mapping tail of run_pair (MACVO_Fast.yaml keeps it on)")
    ap.add_argument("--graph", action="store_true", help="HIP_CUDAGraph_FlowFormerCovFrontend (the network captured as a hipGraph) instead of eager launches")
    ap.add_argument("--channels-last", action="store_true", help="the network's convolution weights in NHWC (torch.channels_last)")
    ap.add_argument("--variants", default="hooked,unhooked")
    args = ap.parse_args()
    assert args.warmup + 2 < args.frames <= 255, "the frame index travels in one 8-bit pixel"
    from tests import refrun

    if refrun.reference_root() is None:
        print(json.dumps({"end_to_end": None, "why": "the reference's Python tree is absent (oracle/_ref/pyref: python oracle/build_ref.py)"}))
        return
    ref = refrun.import_reference()
    import macvo_amd.interfaces as I
    import macvo_amd.plugins  # noqa: F401

    assert I.USING_REFERENCE
    res = {"what": "images -> poses through the reference's unmodified Odometry/MACVO.py loop; learned frontend = FlowFormerCov-shaped network, RANDOM "
                   "weights, PyTorch-ROCm eager (tools/flowformer_host.py) on top of the synthetic scene's fields; selector / covariance / PGO = HIP plugins",
           "config": {"H": args.height, "W": args.width, "enc_dtype": args.enc, "dec_dtype": args.dec, "decoder_depth": args.decoder_depth,
                      "mapping": bool(args.mapping), "frontend": "HIP_CUDAGraph_FlowFormerCovFrontend" if args.graph else "HIP_FlowFormerCovFrontend",
                      "frames": args.frames, "warmup": args.warmup, "channels_last": bool(args.channels_last)},
           "reference_published": "12.5 frames/s, Fast mode, RTX 6000 Ada, trained weights (README.md:28,117)"}
    for v in args.variants.split(","):
        try:
            res[v] = run_variant(ref, refrun, v, args)
        except Exception as e:  # noqa: BLE001
            res[v] = {"error": f"{type(e).__name__}: {e}"}
    if "fps" in res.get("hooked", {}) and "fps" in res.get("unhooked", {}):
        res["hooked_over_unhooked"] = round(res["hooked"]["fps"] / res["unhooked"]["fps"], 3)
        res["hooked_over_unhooked_median"] = round(res["hooked"]["fps_median"] / res["unhooked"]["fps_median"], 3)
    print(json.dumps({"end_to_end": res}))


if __name__ == "__main__":
    main()
