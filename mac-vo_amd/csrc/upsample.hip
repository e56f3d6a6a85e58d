// §8(f) rank 1 — convex 8x upsampling of the 1/8-resolution flow / log-sigma fields (RAFT / FlowFormer `upsample_flow`)
//
// Replaces MemoryDecoder.upsample_flow at its call sites Module/Network/FlowFormerCov/covhead.py:124-126 (flow) and
// :133-135 (log-sigma) — in-tree twin: Module/Network/PWCNet/pwc_cov/gru.py:40-52 — optionally fused with the
// `exp(2 * cov)` of Module/Network/FlowFormerCov/flownet.py:44:
//   mask.view(N, 1, 9, 8, 8, H, W).softmax(2);  unfold(8 * flow, 3x3, pad 1);  sum over the 9 taps;  -> [N, 2, 8H, 8W]
//
// gfx950 design (HBM-bound: 576 mask floats in, 128 floats out per coarse pixel = 2.8 KB/px, 13.6 MB per sample):
//   a wave owns 64 consecutive coarse pixels x one sub-row sy, so every mask channel is one 256-B coalesced load and
//   there are 8 x more waves than pixels/64 to keep ~72 loads per lane in flight; a lane keeps the 3x3 neighbourhood of
//   its pixel in registers and produces the 8 sub-pixels sx = 0..7 of both channels = 32 contiguous bytes per lane per
//   channel -> the wave writes 2 KB contiguous rows with dwordx4 stores.
#include "common.h"
#include <math.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void convex_upsample_kernel(const float* __restrict__ flow,
                                                               const float* __restrict__ mask,
                                                               float* __restrict__ out, int h, int w, float mask_scale,
                                                               int exp2_out) {
    const int b = blockIdx.z;
    const int hw = h * w;
    const int p = blockIdx.x * 64 + threadIdx.x;          // coarse pixel (row-major): one wave = 64 consecutive pixels
    const int sy = blockIdx.y * 4 + threadIdx.y;          // sub-row handled by this wave (8 waves per pixel group)
    if (p >= hw) return;
    const int y = p / w, x = p - y * w;
    const float* fl = flow + (size_t)b * 2 * hw;
    const float* mk = mask + (size_t)b * 576 * hw + p;

    // 3x3 neighbourhood of 8*flow, zero padded (F.unfold(8 * flow, [3,3], padding=1))
    float nb[2][9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        const bool ok = yy >= 0 && yy < h && xx >= 0 && xx < w;
#pragma unroll
        for (int c = 0; c < 2; ++c) nb[c][k] = ok ? 8.f * fl[(size_t)c * hw + yy * w + xx] : 0.f;
    }

    const int W8 = 8 * w;
    {
        float o[2][8];
#pragma unroll
        for (int sx = 0; sx < 8; ++sx) {
            float m[9];
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                m[k] = mask_scale * mk[(size_t)(k * 64 + sy * 8 + sx) * hw];
                mx = fmaxf(mx, m[k]);
            }
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                m[k] = expf(m[k] - mx);
                s += m[k];
            }
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float pk = m[k] / s;
                a0 += pk * nb[0][k];
                a1 += pk * nb[1][k];
            }
            o[0][sx] = exp2_out ? expf(a0 * 2.f) : a0;
            o[1][sx] = exp2_out ? expf(a1 * 2.f) : a1;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float* dst = out + ((size_t)(b * 2 + c) * 8 * h + (8 * y + sy)) * W8 + 8 * x;
            reinterpret_cast<f32x4*>(dst)[0] = f32x4{o[c][0], o[c][1], o[c][2], o[c][3]};
            reinterpret_cast<f32x4*>(dst)[1] = f32x4{o[c][4], o[c][5], o[c][6], o[c][7]};
        }
    }
}

}  // namespace

extern "C" int mv_convex_upsample(const float* flow, const float* mask, float* out, int B, int h, int w,
                                  float mask_scale, int exp2_out, mvStream_t stream) {
    MV_CHECK_ARG(flow && mask && out && B > 0 && h > 0 && w > 0);
    MV_CHECK_ARG(((uintptr_t)out & 15) == 0);
    if (B > 65535) return MV_ERR_UNSUPPORTED;
    dim3 grid(mv_ceil_div(h * w, 64), 2, B), block(64, 4);
    hipLaunchKernelGGL(convex_upsample_kernel, grid, block, 0, (hipStream_t)stream, flow, mask, out, h, w, mask_scale,
                       exp2_out);
    return mv_launch_status();
}
