// Shared device/host helpers for libmacvo_hip (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "macvo_hip.h"

#define MV_WAVE 64

// live rows per lane of a lane-batched launch (kernel argument, passed by value: no H2D copy, no device table)
// Volume epilogue stores: non-temporal.  Measured on MI355X: a store-only kernel in the MFMA C layout writes 184 MB in 28.7 us with
// plain stores and 36.2 us with non-temporal ones (profiles/probes/store_probe.*), but the volume kernels themselves are not bound
// there (fp32: 193 us either way; 16-bit streaming: 41 us either way) and IN THE PIPELINE, where the lookups / selectors of other
// frames run beside the volume, non-temporal stores win because 184 MB per frame do not sweep the L2s: fp32 3.38-3.41 k vs
// 3.31 k frames/s, 16-bit 5.13 k vs 4.83 k, 3 lanes 4.09 k vs 4.05 k (profiles/probes/ab_nt.sh).  -DMV_PLAIN_STORES builds the
// other variant for A/B.
#ifndef MV_PLAIN_STORES
#define MV_VOL_STORE(v, p) __builtin_nontemporal_store((v), (p))
#define MV_VOL_STORE_ASM_MOD " nt"
#else
#define MV_VOL_STORE(v, p) (*(p) = (v))
#define MV_VOL_STORE_ASM_MOD ""
#endif

// Per-pixel frontend epilogue (Frontend.py:183-200, StereoDepth.py:270-282, flownet.py:44), shared by frontend_epilogue_kernel and
// the selector's fused form so that both produce the same bits.  Pointers are already offset to the lane; any output may be null.
struct mvEpiArgs {
    const float* flow;      // [2, 2, H, W]
    const float* logcov;    // [2, 2, H, W]
    int cov_is_log;
    float bl_fx, bl_fx_sq;
    float *disparity, *disparity_cov, *depth, *depth_cov, *match_flow, *match_cov;
    uint8_t* bad_mask;
};
#ifdef __HIPCC__
__device__ __forceinline__ void mv_epilogue_pixel(const mvEpiArgs& a, int plane, int i) {
    // sample 0 (stereo pair): flow[0,0] -> disparity, cov[0,0] -> disparity variance
    const float fx0 = a.flow[i];
    const float lc0 = a.logcov[i];
    const float dcov = a.cov_is_log ? expf(lc0 * 2.f) : lc0;
    const float d = fabsf(fx0);
    if (a.disparity) a.disparity[i] = d;
    if (a.disparity_cov) a.disparity_cov[i] = dcov;
    if (a.depth) a.depth[i] = a.bl_fx * (1.f / d);
    if (a.depth_cov) {
        const float d2 = d * d;
        const float err2 = dcov * (1.f / d2);
        a.depth_cov[i] = a.bl_fx_sq * (err2 / d2);
    }
    if (a.bad_mask) a.bad_mask[i] = fx0 <= 0.f;
    // sample 1 (temporal pair): flow[1] and cov[1] padded with sigma_uv = 0
    if (a.match_flow) {
        a.match_flow[i] = a.flow[2 * plane + i];
        a.match_flow[plane + i] = a.flow[3 * plane + i];
    }
    if (a.match_cov) {
        const float l0 = a.logcov[2 * plane + i], l1 = a.logcov[3 * plane + i];
        a.match_cov[i] = a.cov_is_log ? expf(l0 * 2.f) : l0;
        a.match_cov[plane + i] = a.cov_is_log ? expf(l1 * 2.f) : l1;
        a.match_cov[2 * plane + i] = 0.f;
    }
}
#endif

struct mvLaneCounts {
    int32_t n[MV_MAX_LANES];
};

#define MV_CHECK_ARG(cond) \
    do {                   \
        if (!(cond)) return MV_ERR_INVALID_ARG; \
    } while (0)

// The window lookups of a frame run beside the next frame's volume GEMM (other stream, same CUs).  Raising their wave
// priority lets their few instructions win issue arbitration against the GEMM's MFMA streams: measured 0.3228 -> 0.3168 ms
// per frame with the GEMM's own duration unchanged (219-220 us); raising it for the selector / epilogue kernels as well
// gave the same frame time but stretched the GEMM to 223 us, so only the lookups do it.
#ifndef MV_NO_SMALL_PRIO
#define MV_SMALL_KERNEL_PRIO() __builtin_amdgcn_s_setprio(3)
#else
#define MV_SMALL_KERNEL_PRIO() ((void)0)
#endif

// A/B knob (-DMV_PRIO_CHAIN): the selector / epilogue kernels — the rest of the volume -> lookups -> selector -> host chain whose
// latency bounds a single-sequence stream — at raised wave priority as well
#ifdef MV_PRIO_CHAIN
#define MV_CHAIN_KERNEL_PRIO() __builtin_amdgcn_s_setprio(3)
#else
#define MV_CHAIN_KERNEL_PRIO() ((void)0)
#endif

static inline int mv_launch_status() {
    return hipGetLastError() == hipSuccess ? MV_OK : MV_ERR_LAUNCH;
}

static inline int mv_ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---- wave-level reductions (64 lanes, butterfly so every lane ends with the result) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
