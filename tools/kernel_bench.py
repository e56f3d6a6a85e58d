#!/usr/bin/env python
"""Stand-alone timing of individual C-ABI kernels on the GPU box (HIP events, median of N launches).

    python tools/kernel_bench.py [volume_f32 volume_f16 lookup select cov pgo ...] [--iters 50]

Used for A/B tuning and as the target of `rocprofv3 --pmc ...` runs (one kernel family per process).
"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from macvo_amd import ops  # noqa: E402
from tools import synth  # noqa: E402


def timeit(fn, iters, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return statistics.median(ts), min(ts)


def synth_coords(B, h8, w8, dev):
    g = torch.Generator().manual_seed(1)
    base = torch.stack([torch.arange(w8).float()[None].expand(h8, w8), torch.arange(h8).float()[:, None].expand(h8, w8)])[None]
    return (base + (torch.rand(B, 2, h8, w8, generator=g) * 2 - 1) * 8).to(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["volume_f32", "volume_f16", "lookup", "select", "cov", "pgo"])
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--H", type=int, default=480)
    ap.add_argument("--W", type=int, default=640)
    ap.add_argument("--B", type=int, default=2)
    ap.add_argument("--zeros", action="store_true", help="volume_split: all-zero operands (DVFS probe: same instruction stream, less switching power)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    H, W, B, C = a.H, a.W, a.B, 256
    h8, w8 = H // 8, W // 8
    n = h8 * w8
    g = torch.Generator().manual_seed(0)
    f1, f2 = torch.randn(B, C, h8, w8, generator=g).to(dev), torch.randn(B, C, h8, w8, generator=g).to(dev)
    flops = B * 2.0 * n * n * C
    vol = torch.empty((B * n, 1, h8, w8), dtype=torch.float32, device=dev)
    for w in a.what:
        if w == "volume_f32":
            med, mn = timeit(lambda: ops.corr_volume(f1, f2, "chw", out=vol), a.iters)
            print(f"volume_f32_chw  B={B} {med:8.1f} us (min {mn:.1f})  {flops / med / 1e6:7.1f} TFLOP/s  {flops / med / 1e6 / 157.3 * 100:.1f}% of f32 MFMA peak")
            a1, a2 = f1.permute(0, 2, 3, 1).contiguous(), f2.permute(0, 2, 3, 1).contiguous()
            med, mn = timeit(lambda: ops.corr_volume(a1, a2, "hwc", out=vol), a.iters)
            print(f"volume_f32_hwc  B={B} {med:8.1f} us (min {mn:.1f})  {flops / med / 1e6:7.1f} TFLOP/s")
            med, mn = timeit(lambda: ops.corr_volume(a1, a2, "hwc", out=vol, precision="split3"), a.iters)
            print(f"volume_split3_hwc B={B} {med:6.1f} us (min {mn:.1f})  {flops / med / 1e6:7.1f} TFLOP/s algorithmic")
        elif w == "volume_split":
            z1, z2 = (torch.zeros_like(f1), torch.zeros_like(f2)) if a.zeros else (f1, f2)
            for mode, nprod in (("bf16x3", 6), ("f16x2", 3)):
                if os.environ.get("MV_SPLIT_MODE", mode) != mode:
                    continue
                pk = ops.volume_pack(z1, z2, mode=mode)
                med, mn = timeit(lambda: ops.volume_pack(z1, z2, out=pk, mode=mode), a.iters)
                print(f"volume_pack  {mode} B={B} {med:8.1f} us (min {mn:.1f})")
                for _ in range(100):
                    ops.corr_volume_packed(pk[0], pk[1], B, C, n, n, out=vol, mode=mode)
                med, mn = timeit(lambda: ops.corr_volume_packed(pk[0], pk[1], B, C, n, n, out=vol, mode=mode), a.iters, warm=20)
                print(f"volume_split {mode} B={B} {med:8.1f} us (min {mn:.1f})  {flops / med / 1e6:7.1f} TFLOP/s algorithmic = {nprod * flops / med / 1e6 / 2500 * 100:.1f}% of the "
                      f"16-bit MFMA peak executed ({'zero' if a.zeros else 'random'} operands)")
        elif w == "patch_embed":
            # (f)2: both volumes of a frame (S = B n slices) through the fused conv stack; 2 x (36 + 16*36*... ) MAC per output, see DESIGN
            from tools.synth import patch_embed_weights
            if not ops.cost_patch_embed_supported(h8, w8):
                print(f"patch_embed: no kernel for {h8}x{w8} slices")
                continue
            Wt = [t.to(dev) for t in patch_embed_weights(0)]
            hp, wpd = (h8 + 7) // 8 * 8, (w8 + 7) // 8 * 8
            m1, m2, m3 = hp * wpd // 4, hp * wpd // 16, hp * wpd // 64
            volr = torch.randn(B * n, 1, h8, w8, device=dev) * 16
            outp = torch.empty((B * n, m3, 64), dtype=torch.float32, device=dev)
            fl = B * n * 2.0 * (m1 * 16 * 36 + m2 * 32 * 576 + m3 * 64 * 1152)
            byts = B * n * (h8 * w8 * 4 + m3 * 64 * 4.0)
            print(f"patch_embed: S = {B * n} slices {h8}x{w8} -> {m3} tokens, {fl / 1e9:.1f} GFLOP, MV_PE_STRIP={os.environ.get('MV_PE_STRIP', '0')}")
            for operand in ("f16", "bf16"):
                pk = ops.PatchEmbedWeights(*Wt, operand=operand)
                for _ in range(5):
                    ops.cost_patch_embed(volr, pk, tokens=True, out=outp)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = max(3, a.iters // 5)
                e0.record()
                for _ in range(reps):
                    ops.cost_patch_embed(volr, pk, tokens=True, out=outp)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / reps
                print(f"cost_patch_embed<{operand}> S={B * n} {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s algorithmic = {fl / us / 1e6 / 2500 * 100:.1f}% of the 16-bit MFMA peak; "
                      f"HBM {byts / us / 1e3:.0f} GB/s ({byts / 1e6:.0f} MB)")
                # Fast mode (row (f)2): 16-bit cells in, 16-bit tokens out — half the HBM bytes, no conversion in the staging
                dt16 = torch.float16 if operand == "f16" else torch.bfloat16
                vol16, out16 = volr.to(dt16), torch.empty((B * n, m3, 64), dtype=dt16, device=dev)
                for _ in range(5):
                    ops.cost_patch_embed(vol16, pk, tokens=True, out=out16)
                e0.record()
                for _ in range(reps):
                    ops.cost_patch_embed(vol16, pk, tokens=True, out=out16)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / reps
                print(f"cost_patch_embed<{operand}, 16-bit in/out> S={B * n} {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s algorithmic = {fl / us / 1e6 / 2500 * 100:.1f}% of the 16-bit MFMA "
                      f"peak; HBM {byts / 2 / us / 1e3:.0f} GB/s ({byts / 2e6:.0f} MB)")
                del vol16, out16
            # the unfused form: the same three layers as PyTorch / MIOpen convolutions (bf16, channels_last), intermediates through HBM
            import torch.nn.functional as F
            if (h8, w8) != (60, 80):
                continue                                      # (the unfused chain at 720p is 3 x 10 GB of intermediates: skipped)
            xb = F.pad(volr, (0, 0, 0, 4)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            wb = [t.to(torch.bfloat16) for t in Wt]
            def unfused():
                y = F.relu(F.conv2d(xb, wb[0], wb[1], stride=2, padding=2))
                y = F.relu(F.conv2d(y, wb[2], wb[3], stride=2, padding=2))
                return F.conv2d(y, wb[4], wb[5], stride=2, padding=2)
            try:
                for _ in range(2):
                    unfused()
                e0.record()
                for _ in range(3):
                    unfused()
                e1.record()
                torch.cuda.synchronize()
                print(f"  unfused torch conv2d chain (bf16, channels_last, MIOpen): {e0.elapsed_time(e1) * 1e3 / 3:8.1f} us")
            except Exception as e:  # noqa: BLE001
                print("  unfused torch conv2d chain failed:", repr(e)[:200])
        elif w == "volume_f16":
            for dt in (torch.float16, torch.bfloat16):
                a1, a2 = f1.permute(0, 2, 3, 1).contiguous().to(dt), f2.permute(0, 2, 3, 1).contiguous().to(dt)
                byts = B * (2.0 * n * C * 2 + 4.0 * n * n)
                med, mn = timeit(lambda: ops.corr_volume(a1, a2, "hwc", out=vol), a.iters)
                print(f"volume_{str(dt)[6:]}_hwc B={B} {med:8.1f} us (min {mn:.1f})  {byts / med / 1e3:7.1f} GB/s  {byts / med / 1e3 / 8000 * 100:.1f}% of HBM peak")
                v16 = ops.corr_volume_out16(a1, a2)
                if v16 is not None:
                    b16 = B * (2.0 * n * C * 2 + 2.0 * n * n)
                    med, mn = timeit(lambda: ops.corr_volume_out16(a1, a2, out=v16), a.iters)
                    print(f"volume_{str(dt)[6:]}_hwc_out16 B={B} {med:8.1f} us (min {mn:.1f})  {b16 / med / 1e3:7.1f} GB/s  {b16 / med / 1e3 / 8000 * 100:.1f}% of HBM peak ({b16 / 1e6:.0f} MB)")
                    if dt == torch.float16:
                        co = synth_coords(B, h8, w8, dev)
                        tok = ops.corr_lookup(v16, co, 4)
                        med, mn = timeit(lambda: ops.corr_lookup(v16, co, 4, out=tok), a.iters)
                        print(f"lookup_vol16 B={B} {med:8.1f} us (min {mn:.1f})")
                        vt = ops.corr_volume_out16(a1, a2, tiled=True)
                        med, mn = timeit(lambda: ops.corr_lookup(vt, co, 4, out=tok, tiled=True, image_hw=(h8, w8)), a.iters)
                        print(f"lookup_vol16_tiled B={B} {med:8.1f} us (min {mn:.1f})")
                        del vt
                        v32 = v16.float()
                        med, mn = timeit(lambda: ops.corr_lookup(v32, co, 4, out=tok), a.iters)
                        print(f"lookup_vol32 B={B} {med:8.1f} us (min {mn:.1f})")
                        del v32
                    del v16
                c1, c2 = f1.to(dt), f2.to(dt)
                med, mn = timeit(lambda: ops.corr_volume(c1, c2, "chw", out=vol), a.iters)
                print(f"volume_{str(dt)[6:]}_chw B={B} {med:8.1f} us (min {mn:.1f})  {byts / med / 1e3:7.1f} GB/s")
        elif w == "lookup":
            ops.corr_volume(f1, f2, "chw", out=vol)
            from tools.synth import coords_grid

            coords = (coords_grid(B, h8, w8) + (torch.rand(B, 2, h8, w8, generator=g) * 2 - 1) * 8).to(dev)
            tok = torch.empty((B, 81, h8, w8), dtype=torch.float32, device=dev)
            byts = B * (n * 100 * 4 + n * 8 + n * 81 * 4.0)
            med, mn = timeit(lambda: ops.corr_lookup(vol, coords, 4, out=tok), a.iters)
            print(f"lookup r=4      B={B} {med:8.1f} us (min {mn:.1f})  {byts / med / 1e3:7.1f} GB/s algorithmic")
        elif w == "localcorr":
            # PWC pyramid levels of a 640x448 input (pwc_model.py:178-233): (C, H, W)
            for (cc, hh, ww) in ((196, 7, 10), (128, 14, 20), (96, 28, 40), (64, 56, 80), (32, 112, 160)):
                a1, a2 = torch.randn(1, cc, hh, ww, generator=g).to(dev), torch.randn(1, cc, hh, ww, generator=g).to(dev)
                o = torch.empty(1, 81, hh, ww, device=dev)
                med, mn = timeit(lambda: ops.local_corr81(a1, a2, out=o), a.iters)
                fl = 2.0 * 81 * cc * hh * ww
                print(f"local_corr81 C={cc:3d} {hh}x{ww:<3d} {med:7.1f} us (min {mn:.1f})  {fl / med / 1e3:8.1f} GFLOP/s")
        elif w == "upsample":
            fl = torch.randn(B, 2, h8, w8, generator=g).to(dev)
            mk = torch.randn(B, 576, h8, w8, generator=g).to(dev)
            byts = B * (578 * n * 4 + 2 * 64 * n * 4.0)
            for mdt in (torch.float32, torch.bfloat16):
                mkd = mk.to(mdt)
                byts = B * ((576 * mkd.element_size() + 8) * n + 2 * 64 * n * 4.0)
                for _ in range(5):
                    ops.convex_upsample(fl, mkd, 0.25)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 4 * a.iters
                e0.record()
                for _ in range(reps):          # back to back: the figure is the kernel's period, not a launch + event round trip
                    ops.convex_upsample(fl, mkd, 0.25)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / reps
                print(f"convex_upsample B={B} mask {str(mdt)[6:]} {us:8.2f} us back to back  {byts / us / 1e3:7.1f} GB/s  {byts / us / 1e3 / 8000 * 100:.1f}% of HBM peak ({byts / 1e6:.1f} MB)")
        elif w == "select":
            fc = synth.flow_cov_maps(H, W, 2).to(dev)
            d0, d0c = [t.to(dev) for t in synth.depth_maps(H, W, 3)]
            d1, d1c = [t.to(dev) for t in synth.depth_maps(H, W, 4)]
            med, mn = timeit(lambda: ops.kp_select("nodepth", H, W, flow_cov=fc, kernel_size=7, mask_width=32, max_match_cov=100.0), a.iters)
            print(f"kp_select nodepth    {med:8.1f} us (min {mn:.1f})")
            med, mn = timeit(lambda: ops.kp_select("full", H, W, flow_cov=fc, depth0=d0, depth0_cov=d0c, depth1=d1, depth1_cov=d1c,
                                                   kernel_size=7, mask_width=32, max_depth=80.0, max_depth_cov=250.0, max_match_cov=100.0), a.iters)
            print(f"kp_select full       {med:8.1f} us (min {mn:.1f})")
            med, mn = timeit(lambda: ops.kp_select("mapping", H, W, depth0=d0, depth0_cov=d0c, mask_width=32, max_depth=20.0, max_depth_cov=0.2), a.iters)
            print(f"kp_select mapping    {med:8.1f} us (min {mn:.1f})")
        elif w == "cov":
            depth = synth.depth_maps(H, W, 3)[0].to(dev)
            for npt in (200, 2000):
                kp = synth.keypoints(npt, H, W, 5).float().to(dev)
                fc = (torch.ones(npt, 3) * 0.25).to(dev)
                med, mn = timeit(lambda: ops.match_cov(depth, kp, fc, None, 320.0, 320.0, 320.0, 240.0), a.iters)
                print(f"match_cov N={npt:5d}    {med:8.1f} us (min {mn:.1f})")
        elif w == "pgo":
            from tools.synth import pgo_batch as _to_batch, pgo_problem

            for nprob in (1, 8, 256, 4096):
                base = [pgo_problem(n=200, seed=6 + k)[0] for k in range(min(nprob, 8))]
                probs = [base[k % len(base)] for k in range(nprob)]
                batch = _to_batch(probs, dev)
                for gt in ("disp", "icp"):
                    med, mn = timeit(lambda: ops.pgo_solve(batch, gt), max(5, a.iters // 5))
                    print(f"pgo {gt:6s} nprob={nprob:5d} {med:9.1f} us (min {mn:.1f})  {nprob / med * 1e6:10.0f} solves/s")


if __name__ == "__main__":
    main()
