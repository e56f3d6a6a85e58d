// §8(f) rank 1 — convex 8x upsampling of the 1/8-resolution flow / log-sigma fields (RAFT / FlowFormer `upsample_flow`)
//
// Replaces MemoryDecoder.upsample_flow at its call sites Module/Network/FlowFormerCov/covhead.py:124-126 (flow) and
// :133-135 (log-sigma) — in-tree twin: Module/Network/PWCNet/pwc_cov/gru.py:40-52 — optionally fused with the
// `exp(2 * cov)` of Module/Network/FlowFormerCov/flownet.py:44:
//   mask.view(N, 1, 9, 8, 8, H, W).softmax(2);  unfold(8 * flow, 3x3, pad 1);  sum over the 9 taps;  -> [N, 2, 8H, 8W]
//
// gfx950 design.  HBM-bound: 576 mask values in, 128 floats out per coarse pixel (2.8 KB/px with an fp32 mask, 13.6 MB per
// 60x80 sample), and at these sizes (22 MB per call) the whole transfer has to be IN FLIGHT at once to approach HBM speed:
// one round trip is ~2 us, the transfer itself ~3 us.  So:
//   * a wave owns 64 x PX consecutive coarse pixels x one sub-row sy x NSX sub-columns; every mask channel is one coalesced
//     256-B load per wave, and ALL 9 x NSX of them are issued before the first is consumed (registers, no LDS: nothing is
//     shared between lanes);
//   * the 3x3 neighbourhood of 8*flow is loaded branch-free (clamped address + select) so its 18 loads are in flight with
//     the mask loads instead of 18 serial round trips behind zero-padding branches (the round-1 kernel: 23-26 us);
//   * no early exit: tail lanes read clamped addresses and skip their stores;
//   * softmax weights with ONE division per sub-pixel (r = 1/sum, then 9 multiplies) instead of nine;
//   * a lane produces NSX contiguous floats per channel -> dwordx4 stores, a wave writes whole 2-KB row segments;
//   * 16-bit masks (the decoder's autocast type in Fast mode, MACVO_Fast.yaml:73-74) are read as they are (16-bit loads, one
//     pixel per lane) and widened in registers: the arithmetic is the fp32 formula on the widened values, no 11 -> 22 MB
//     `.float()` pass in front.  (After the exp fix the kernel is bound by its ~850 VALU instructions per wave and by one
//     memory round trip, not by bytes: the 16-bit form takes the same 7.4 us as the fp32 one.)
#include "common.h"
#include <math.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int DT> struct MaskElem;
template <> struct MaskElem<MV_F32> { typedef float T; static __device__ __forceinline__ float widen(float v) { return v; } };
template <> struct MaskElem<MV_F16> {
    typedef uint16_t T;
    static __device__ __forceinline__ float widen(uint16_t v) {
        _Float16 hv;
        __builtin_memcpy(&hv, &v, 2);
        return (float)hv;
    }
};
template <> struct MaskElem<MV_BF16> {
    typedef uint16_t T;
    static __device__ __forceinline__ float widen(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
};

template <typename T, int PX> struct PixVec { T v[PX]; };

// exp(x) for the softmax numerators (x = m - max <= 0: no overflow; a weight below 2^-126 becomes 0, as it does in any fp32 softmax to
// 1e-38 of the sum).  The library expf is ~25 instructions (range reduction + overflow / denormal selects) and 36-72 of them per wave made the
// kernel VALU-bound (1238 instructions per wave for 54 loads).  Here: t = x log2(e) with its rounding error recovered by one fma
// (lo = fma(x, L2E, -t) + x * (log2(e) - L2E)), 2^t by the hardware's v_exp_f32 (1 ulp over its whole range), 2^lo = 1 + lo ln 2 to
// first order (|lo| < 2^-17 for |x| < 128: the dropped term is < 2^-36): ~1.5 ulp, 6 instructions.
// x = -inf (an fp16 mask head that overflowed under autocast) or x log2(e) overflowing would make `lo` inf - inf = NaN where expf / torch.softmax give the
// tap weight 0: arguments below -126 are clamped there — 2^(-126 log2 e) is 0 in fp32 as well; a NaN argument stays NaN (compare + select, not max).
__device__ __forceinline__ float exp_nonpos(float x) {
    const float L2E = 1.44269502f, L2E_LO = 1.92596303e-8f, LN2 = 0.693147182f;
    x = x < -126.f ? -126.f : x;
    const float t = x * L2E;
    const float lo = __builtin_fmaf(x, L2E, -t) + x * L2E_LO;
    const float e = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(e, lo * LN2, e);
}

// grid: x = groups of 64*PX coarse pixels, y = (8 sub-rows x 8/NSX sub-column groups) / 4 waves, z = sample
template <int DT, int PX, int NSX>
__global__ __launch_bounds__(256) void convex_upsample_kernel(const float* __restrict__ flow,
                                                               const typename MaskElem<DT>::T* __restrict__ mask,
                                                               float* __restrict__ out, int h, int w, float mask_scale,
                                                               int exp2_out) {
    typedef typename MaskElem<DT>::T T;
    typedef PixVec<T, PX> V;
    constexpr int SPLIT = 8 / NSX;
    const int b = blockIdx.z;
    const int hw = h * w;
    const int wv = blockIdx.y * 4 + threadIdx.y;
    const int sy = wv / SPLIT, sx0 = (wv % SPLIT) * NSX;
    const int p0 = (blockIdx.x * 64 + threadIdx.x) * PX;   // first coarse pixel of this lane (row-major)
    const int pl = min(p0, hw - PX);                        // tail lanes: clamped loads, no stores

    // every mask value this lane needs: 9 taps x NSX sub-columns (x PX pixels), issued back to back
    const T* mk = mask + (size_t)b * 576 * hw + (size_t)(sy * 8 + sx0) * hw + pl;
    V raw[9][NSX];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int sx = 0; sx < NSX; ++sx) raw[k][sx] = *reinterpret_cast<const V*>(mk + (size_t)(k * 64 + sx) * hw);

    // 3x3 neighbourhood of 8*flow, zero padded (F.unfold(8 * flow, [3,3], padding=1)): clamped address + select
    const float* fl = flow + (size_t)b * 2 * hw;
    int py[PX], px[PX];
    float nb[PX][2][9];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        py[j] = (pl + j) / w;
        px[j] = (pl + j) - py[j] * w;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = py[j] + k / 3 - 1, xx = px[j] + k % 3 - 1;
            const bool ok = yy >= 0 && yy < h && xx >= 0 && xx < w;
            const int idx = min(max(yy, 0), h - 1) * w + min(max(xx, 0), w - 1);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float v = fl[(size_t)c * hw + idx];
                nb[j][c][k] = ok ? 8.f * v : 0.f;
            }
        }
    }

    // everything above is issued before anything below is consumed: the machine scheduler would otherwise sink each load next to
    // its use (fewer live registers, one round trip per sub-pixel)
    __builtin_amdgcn_sched_barrier(0);

    const int W8 = 8 * w;
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        float o[2][NSX];
#pragma unroll
        for (int sx = 0; sx < NSX; ++sx) {
            float m[9];
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                m[k] = mask_scale * MaskElem<DT>::widen(raw[k][sx].v[j]);
                mx = fmaxf(mx, m[k]);
            }
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                m[k] = exp_nonpos(m[k] - mx);
                s += m[k];
            }
            const float r = 1.f / s;
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float pk = m[k] * r;
                a0 += pk * nb[j][0][k];
                a1 += pk * nb[j][1][k];
            }
            o[0][sx] = exp2_out ? expf(a0 * 2.f) : a0;
            o[1][sx] = exp2_out ? expf(a1 * 2.f) : a1;
        }
        if (p0 + j < hw) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float* dst = out + ((size_t)(b * 2 + c) * 8 * h + (8 * py[j] + sy)) * W8 + 8 * px[j] + sx0;
                if constexpr (NSX == 2) {
                    *reinterpret_cast<f32x2*>(dst) = f32x2{o[c][0], o[c][1]};
                } else {
#pragma unroll
                    for (int q = 0; q < NSX / 4; ++q)
                        reinterpret_cast<f32x4*>(dst)[q] = f32x4{o[c][4 * q], o[c][4 * q + 1], o[c][4 * q + 2], o[c][4 * q + 3]};
                }
            }
        }
    }
}

template <int DT, int PX, int NSX>
int launch(const float* flow, const void* mask, float* out, int B, int h, int w, float mask_scale, int exp2_out,
           hipStream_t stream) {
    dim3 grid(mv_ceil_div(h * w, 64 * PX), 8 * (8 / NSX) / 4, B), block(64, 4);
    hipLaunchKernelGGL((convex_upsample_kernel<DT, PX, NSX>), grid, block, 0, stream, flow,
                       (const typename MaskElem<DT>::T*)mask, out, h, w, mask_scale, exp2_out);
    return mv_launch_status();
}

// Measured on MI355X (profiles/r05_upsample_ab.log; B = 2 fields, back-to-back launches):
//   fp32 mask   60x80: NSX 2 / 4 / 8 = 7.1 / 7.4 / 8.1 us;  90x160: 16.8 / 18.9 / 18.2 us    -> NSX = 2 (4800 waves at 60x80: 4.7 per SIMD)
//   16-bit mask 60x80: NSX 2 / 4 = 7.4-7.8 / 7.4 us;        90x160: 15.4-16.8 / 15.1-15.6 us -> NSX = 4
//   two pixels per lane with dword loads of a 16-bit mask (half the waves, the same loads per wave): 9.3-9.5 us / 17.5-21.9 us -> not built in
template <int DT>
int dispatch(const float* flow, const void* mask, float* out, int B, int h, int w, float mask_scale, int exp2_out,
             hipStream_t stream) {
    if constexpr (DT == MV_F32) return launch<DT, 1, 2>(flow, mask, out, B, h, w, mask_scale, exp2_out, stream);
    else return launch<DT, 1, 4>(flow, mask, out, B, h, w, mask_scale, exp2_out, stream);
}

}  // namespace

extern "C" int mv_convex_upsample_m(const float* flow, const void* mask, int mask_dtype, float* out, int B, int h, int w,
                                    float mask_scale, int exp2_out, mvStream_t stream) {
    MV_CHECK_ARG(flow && mask && out && B > 0 && h > 0 && w > 0);
    MV_CHECK_ARG(((uintptr_t)out & 15) == 0);
    MV_CHECK_ARG(((uintptr_t)mask & (mask_dtype == MV_F32 ? 3 : 1)) == 0);
    if (B > 65535) return MV_ERR_UNSUPPORTED;
    switch (mask_dtype) {
        case MV_F32: return dispatch<MV_F32>(flow, mask, out, B, h, w, mask_scale, exp2_out, (hipStream_t)stream);
        case MV_F16: return dispatch<MV_F16>(flow, mask, out, B, h, w, mask_scale, exp2_out, (hipStream_t)stream);
        case MV_BF16: return dispatch<MV_BF16>(flow, mask, out, B, h, w, mask_scale, exp2_out, (hipStream_t)stream);
        default: return MV_ERR_UNSUPPORTED;
    }
}

extern "C" int mv_convex_upsample(const float* flow, const float* mask, float* out, int B, int h, int w,
                                  float mask_scale, int exp2_out, mvStream_t stream) {
    return mv_convex_upsample_m(flow, mask, MV_F32, out, B, h, w, mask_scale, exp2_out, stream);
}
