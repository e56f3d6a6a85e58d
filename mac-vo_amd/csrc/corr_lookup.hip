// A6 — (2r+1)^2 bilinear window lookup in each query pixel's own H2 x W2 cost slice
//
// Replaces FlowFormer MemoryDecoder.encode_flow_token(cost_maps, coords1) called at
// Module/Network/FlowFormerCov/covhead.py:92 ("MUST run in fp32", :91): RAFT window
// delta[i][j] = (dy[i], dx[j]) added to (x, y)  =>  channel k = K*i + j samples (x + i - r, y + j - r),
// bilinear_sampler = grid_sample(align_corners=True, zeros) after normalising by (W2-1), (H2-1).
//
// gfx950 design (HBM/L2 gather-bound, ~0.7 KB useful per query)
//   * a 256-thread workgroup owns 32 consecutive queries of one batch item; each wave owns 8 of them.
//   * staging: the wave gathers, for its 8 queries, the (K+3)^2 cell block that covers every tap
//     (block origin floor(x)-r-1: one spare cell each side absorbs the fp32 normalise/un-normalise
//     round trip) with all 18 wave-wide loads in flight at once (row segments of 48 B per query),
//     zero-filling out-of-image cells (= zero padding) -> LDS.
//   * compute: lane = tap (81 of 128... of 64x2), per-tap fp32 arithmetic replays grid_sample's exactly
//     (true division, same op order) so results match the ATen CPU kernel to rounding.
//   * output [B, K*K, H1, W1] is channel-major: results are transposed through LDS so every channel row
//     is written as one 128-B segment of 32 consecutive queries.
#include "common.h"

namespace {

template <int R, int QPW>
__global__ __launch_bounds__(64 * (32 / QPW)) void corr_lookup_kernel(const float* __restrict__ vol,
                                                           const float* __restrict__ coords,
                                                           float* __restrict__ out, int N1, int H2, int W2) {
    constexpr int K = 2 * R + 1;
    constexpr int KK = K * K;
    constexpr int BS = K + 3;            // staged block edge (12 for r = 4)
    constexpr int CELLS = BS * BS;       // 144
    constexpr int QPB = 32;              // queries per workgroup (one 128-B output segment per channel)
    constexpr int NWAVE = QPB / QPW;     // waves per workgroup, QPW queries each
    constexpr int NTHR = 64 * NWAVE;
    constexpr int NLOAD = (QPW * CELLS + 63) / 64;  // 18 wave-wide loads
    constexpr int TAP_ROUNDS = (KK + 63) / 64;      // 2 for r = 4

    // one LDS region, two lives: staged cell blocks (read by the tap phase), then the transposed outputs.  Keeping the
    // footprint at max(22.5, 10.7) KB lets a lookup workgroup co-reside with four 32-KB volume-GEMM workgroups on a CU
    // (the pipeline runs the next frame's GEMM on another stream while this frame's lookups execute).
    constexpr int BLK_FLOATS = NWAVE * (QPW * CELLS + 64);
    constexpr int OUT_FLOATS = KK * (QPB + 1);
    __shared__ float smem[BLK_FLOATS > OUT_FLOATS ? BLK_FLOATS : OUT_FLOATS];
    float (*blk)[QPW * CELLS + 64] = reinterpret_cast<float (*)[QPW * CELLS + 64]>(smem);
    float (*outs)[QPB + 1] = reinterpret_cast<float (*)[QPB + 1]>(smem);

    const int b = blockIdx.y;
    const int q0 = blockIdx.x * QPB;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int slice = H2 * W2;

    // lane s < 8 owns query s of this wave: load its coords, derive the block origin
    const int qmine = q0 + wave * QPW + (lane & (QPW - 1));
    const bool qvalid = qmine < N1;
    float x = 0.f, y = 0.f;
    if (qvalid) {
        x = coords[((size_t)b * 2 + 0) * N1 + qmine];
        y = coords[((size_t)b * 2 + 1) * N1 + qmine];
    }
    // clamp only the integer origin (NaN / huge coords must not produce wild addresses)
    const float xc = fminf(fmaxf(x, -1.0e6f), 1.0e6f), yc = fminf(fmaxf(y, -1.0e6f), 1.0e6f);
    const int bx = ((xc == xc) ? (int)floorf(xc) : 0) - R - 1;
    const int by = ((yc == yc) ? (int)floorf(yc) : 0) - R - 1;

    // ---- stage 8 x 144 cells: issue every load before touching LDS
    float v[NLOAD];
#pragma unroll
    for (int r = 0; r < NLOAD; ++r) {
        const int idx = r * 64 + lane;
        const int s = idx / CELLS;
        const int c = idx - s * CELLS;
        const int cy = c / BS, cx = c - cy * BS;
        const int sbx = __shfl(bx, s & (QPW - 1), 64), sby = __shfl(by, s & (QPW - 1), 64);
        const int q = q0 + wave * QPW + s;
        const int gx = sbx + cx, gy = sby + cy;
        const bool ok = (s < QPW) && (q < N1) && gx >= 0 && gx < W2 && gy >= 0 && gy < H2;
        v[r] = ok ? vol[((size_t)b * N1 + q) * slice + gy * W2 + gx] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < NLOAD; ++r) blk[wave][r * 64 + lane] = v[r];
    __syncthreads();

    // ---- taps: lane -> (i, j); replay RAFT normalise + ATen unnormalise + bilinear weights in fp32
    const float wm1 = (float)(W2 - 1), hm1 = (float)(H2 - 1);
    const float sx = wm1 / 2.f, sy = hm1 / 2.f;
    float res[TAP_ROUNDS][QPW];
#pragma unroll
    for (int tr = 0; tr < TAP_ROUNDS; ++tr) {
        const int tap = tr * 64 + lane;
        const int ti = tap / K, tj = tap - ti * K;
#pragma unroll
        for (int s = 0; s < QPW; ++s) {
            const float qx = __shfl(x, s, 64), qy = __shfl(y, s, 64);
            const int sbx = __shfl(bx, s, 64), sby = __shfl(by, s, 64);
            if (tap < KK) {
                const float xs = qx + (float)(ti - R);
                const float ys = qy + (float)(tj - R);
                const float xg = (2.f * xs) / wm1 - 1.f;
                const float yg = (2.f * ys) / hm1 - 1.f;
                const float ix = (xg + 1.f) * sx;
                const float iy = (yg + 1.f) * sy;
                const float fx0 = floorf(ix), fy0 = floorf(iy);
                const float w = ix - fx0, e = 1.f - w;
                const float n = iy - fy0, so = 1.f - n;
                int cx = (int)fminf(fmaxf(fx0, -2.0e6f), 2.0e6f) - sbx;
                int cy = (int)fminf(fmaxf(fy0, -2.0e6f), 2.0e6f) - sby;
                const bool inblk = cx >= 0 && cx <= BS - 2 && cy >= 0 && cy <= BS - 2;
                cx = min(max(cx, 0), BS - 2);
                cy = min(max(cy, 0), BS - 2);
                const float* p = &blk[wave][s * CELLS + cy * BS + cx];
                const float vnw = inblk ? p[0] : 0.f, vne = inblk ? p[1] : 0.f;
                const float vsw = inblk ? p[BS] : 0.f, vse = inblk ? p[BS + 1] : 0.f;
                // ATen: (nw_val*nw + ne_val*ne) + sw_val*sw + se_val*se with nw = s*e, ne = s*w, sw = n*e, se = n*w
                float r0 = vnw * (so * e);
                r0 = r0 + vne * (so * w);
                r0 = r0 + vsw * (n * e);
                r0 = r0 + vse * (n * w);
                res[tr][s] = r0;
            }
        }
    }
    __syncthreads();   // every wave is done reading the staged blocks: the region becomes the output transpose buffer
#pragma unroll
    for (int tr = 0; tr < TAP_ROUNDS; ++tr) {
        const int tap = tr * 64 + lane;
        if (tap < KK) {
#pragma unroll
            for (int s = 0; s < QPW; ++s) outs[tap][wave * QPW + s] = res[tr][s];
        }
    }
    __syncthreads();

    // ---- transposed store: each channel row = 32 consecutive queries (128 B)
    for (int idx = t; idx < KK * QPB; idx += NTHR) {
        const int k = idx >> 5, c = idx & 31;
        if (q0 + c < N1) out[((size_t)b * KK + k) * N1 + q0 + c] = outs[k][c];
    }
}

}  // namespace

// queries-per-launch at or below which the 16-wave variant is used; MV_LOOKUP_SMALL=<n> overrides it for A/B runs
// (tools/kernel_bench.py), e.g. 0 forces the 4-wave variant
#include <stdlib.h>
static int lookup_small_threshold() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("MV_LOOKUP_SMALL");
        v = e ? atoi(e) : 65536;
    }
    return v;
}

extern "C" int mv_corr_lookup(const float* vol, const float* coords, float* out, int B, int H1, int W1,
                              int H2, int W2, int radius, mvStream_t stream) {
    MV_CHECK_ARG(vol && coords && out);
    MV_CHECK_ARG(B > 0 && H1 > 0 && W1 > 0 && H2 > 1 && W2 > 1);
    if (radius < 1 || radius > 4) return MV_ERR_UNSUPPORTED;
    if (B > 65535) return MV_ERR_UNSUPPORTED;
    const int N1 = H1 * W1;
    dim3 grid(mv_ceil_div(N1, 32), B);
    hipStream_t s = (hipStream_t)stream;
    // small launches (one frame: ~300 workgroups for 256 CUs) are latency-bound -> spread a workgroup's 32 queries
    // over 16 waves; large batches are throughput-bound -> 8 queries per wave amortise the per-wave setup
    const bool small = (size_t)B * N1 <= (size_t)lookup_small_threshold();
#define MV_LOOKUP(R)                                                                                                  \
    if (small)                                                                                                        \
        hipLaunchKernelGGL((corr_lookup_kernel<R, 2>), grid, dim3(1024), 0, s, vol, coords, out, N1, H2, W2);         \
    else                                                                                                              \
        hipLaunchKernelGGL((corr_lookup_kernel<R, 8>), grid, dim3(256), 0, s, vol, coords, out, N1, H2, W2)
    switch (radius) {
        case 1: MV_LOOKUP(1); break;
        case 2: MV_LOOKUP(2); break;
        case 3: MV_LOOKUP(3); break;
        default: MV_LOOKUP(4); break;
    }
#undef MV_LOOKUP
    return mv_launch_status();
}
