// A23 — PWC-Net 81-channel local correlation (the reference's only hand-written CUDA kernel)
//
// Replaces Module/Network/PWCNet/pwc/correlation.py forward: kernel_Correlation_rearrange :8-33 (NCHW -> zero-padded
// NHWC copies) + kernel_Correlation_updateOutput :35-103 (one 32-thread block per output pixel, 81 dot products over C,
// strided partial sums + serial reduce by lane 0), launched by _FunctionCorrelation.forward :277-325:
//     out[b, 9*(dy+4) + (dx+4), y, x] = (1/C) * sum_c first[b,c,y,x] * second[b,c,y+dy,x+dx],   dx,dy in [-4,4],
// zero outside the image.  Users: PWCDCNet (pwc_model.py:178-233: five pyramid levels, C = 196..32, 1/64..1/4
// resolution), pwc_model_tartanvo.py:231,259, RAFTCov.py.  The reference has no CPU path (raise NotImplementedError,
// :323-324) and the gradient kernels (:105-233) are training-only: forward only here.
//
// gfx950 design.  The largest call of the network (160x112 pixels, C = 32) is 46 MFLOP and 2.3 MB of input: launch /
// latency bound.  No padded NHWC copies (two launches and 2x the traffic in the reference): NCHW is read directly,
//   * a 256-thread workgroup owns a 16x16 pixel tile and ONE THIRD of the displacement rows (3 dy x 9 dx = 27 outputs
//     per pixel): 3x more workgroups for the tiny pyramid levels, 27 accumulators per thread;
//   * per chunk of 8 channels the [8][18][24] window of `second` (tile + halo, zero filled outside the image) is staged
//     in LDS with coalesced row loads; `first` is read straight into registers (one value per thread per channel);
//   * thread = pixel: 27 LDS reads + 27 fmaf per channel (row stride 25 floats: conflict-free), channels accumulate in
//     order c = 0..C-1 in fp32, then the reference's division by C;
//   * stores are channel-major rows of 16 consecutive pixels (64-B segments).
// The coarse pyramid levels are the opposite shape — 70 pixels x 196 channels at 1/64 resolution: three tiles' worth of
// pixels and a 25-chunk sequential channel loop (measured 108 us).  They take the DIRECT kernel instead: one thread per
// output element x S-way channel split (lane = (channel group, x)), every operand read straight from L1/L2 (the whole
// input is 55-440 KB), 8 channel steps in flight per thread, S partial sums combined with wave shuffles.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int LC_T = 16;          // tile edge (pixels)
constexpr int LC_MD = 4;          // max displacement
constexpr int LC_CK = 8;          // channels staged per pass
constexpr int LC_ROWS = LC_T + 2; // window rows for 3 displacement rows
constexpr int LC_COLS = LC_T + 2 * LC_MD;
constexpr int LC_LD = LC_COLS + 1;

__global__ __launch_bounds__(256) void local_corr81_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                            float* __restrict__ out, int C, int H, int W) {
    __shared__ float win[LC_CK][LC_ROWS][LC_LD];
    const int tx = threadIdx.x & (LC_T - 1), ty = threadIdx.x >> 4;
    const int b = blockIdx.z / 3, g = blockIdx.z - 3 * b;   // g: displacement rows dy = 3g-4 .. 3g-2
    const int x0 = blockIdx.x * LC_T, y0 = blockIdx.y * LC_T;
    const int x = x0 + tx, y = y0 + ty;
    const bool inside = x < W && y < H;
    const size_t plane = (size_t)H * W;
    const float* p1 = f1 + (size_t)b * C * plane;
    const float* p2 = f2 + (size_t)b * C * plane;
    const int wy0 = y0 + 3 * g - LC_MD, wx0 = x0 - LC_MD;   // window origin in the image

    float acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0.f;

    for (int c0 = 0; c0 < C; c0 += LC_CK) {
        const int nc = min(LC_CK, C - c0);
        float a[LC_CK];
#pragma unroll
        for (int c = 0; c < LC_CK; ++c) a[c] = (inside && c < nc) ? p1[(size_t)(c0 + c) * plane + (size_t)y * W + x] : 0.f;
        __syncthreads();   // previous chunk fully consumed
        constexpr int NST = (LC_CK * LC_ROWS * LC_COLS + 255) / 256;   // 14 staged values per thread, all loads in flight
        float st[NST];
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int e = i * 256 + (int)threadIdx.x;
            const int c = e / (LC_ROWS * LC_COLS), r = e - c * (LC_ROWS * LC_COLS);
            const int wy = r / LC_COLS, wx = r - wy * LC_COLS;
            const int gy = wy0 + wy, gx = wx0 + wx;
            st[i] = 0.f;
            if (c < nc && gy >= 0 && gy < H && gx >= 0 && gx < W) st[i] = p2[(size_t)(c0 + c) * plane + (size_t)gy * W + gx];
        }
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int e = i * 256 + (int)threadIdx.x;
            const int c = e / (LC_ROWS * LC_COLS), r = e - c * (LC_ROWS * LC_COLS);
            const int wy = r / LC_COLS, wx = r - wy * LC_COLS;
            if (e < LC_CK * LC_ROWS * LC_COLS) win[c][wy][wx] = st[i];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < LC_CK; ++c) {
            if (c < nc) {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 9; ++dx) acc[dy * 9 + dx] = fmaf(a[c], win[c][ty + dy][tx + dx], acc[dy * 9 + dx]);
            }
        }
    }
    if (inside) {
        const float cf = (float)C;
        float* o = out + ((size_t)b * 81 + 27 * g) * plane + (size_t)y * W + x;
#pragma unroll
        for (int k = 0; k < 27; ++k) o[(size_t)k * plane] = acc[k] / cf;
    }
}

// thread = (output element, channel group s of S); a wave covers 64/S consecutive output elements (x fastest)
template <int S>
__global__ __launch_bounds__(256) void local_corr81_direct(const float* __restrict__ f1, const float* __restrict__ f2,
                                                            float* __restrict__ out, int C, int H, int W, long total) {
    constexpr int XL = 64 / S;
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long o = wave * XL + (lane % XL);
    const int s = lane / XL;
    const bool valid = o < total;
    const long oo = valid ? o : 0;
    const int x = (int)(oo % W);
    long t = oo / W;
    const int y = (int)(t % H);
    t /= H;
    const int k = (int)(t % 81);
    const int b = (int)(t / 81);
    const int x2 = x + k % 9 - LC_MD, y2 = y + k / 9 - LC_MD;
    const bool inb = valid && x2 >= 0 && x2 < W && y2 >= 0 && y2 < H;
    const size_t plane = (size_t)H * W;
    float acc = 0.f;
    if (inb) {
        const float* p1 = f1 + ((size_t)b * C + s) * plane + (size_t)y * W + x;
        const float* p2 = f2 + ((size_t)b * C + s) * plane + (size_t)y2 * W + x2;
        const size_t step = plane * S;
#pragma unroll 8
        for (int c = s; c < C; c += S) {
            acc = fmaf(*p1, *p2, acc);
            p1 += step;
            p2 += step;
        }
    }
#pragma unroll
    for (int off = XL; off < 64; off <<= 1) acc += __shfl_xor(acc, off, 64);
    if (s == 0 && valid) out[o] = acc / (float)C;
}

}  // namespace

extern "C" int mv_local_corr81(const float* first, const float* second, float* out, int B, int C, int H, int W,
                               mvStream_t stream) {
    MV_CHECK_ARG(first && second && out);
    MV_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0);
    if ((size_t)B * 3 > 65535) return MV_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * 81 * H * W;
    static int tiled_min = -1;   // planes at or above this many pixels take the LDS-tiled kernel (MV_LOCALCORR_TILED_MIN)
    if (tiled_min < 0) { const char* e = getenv("MV_LOCALCORR_TILED_MIN"); tiled_min = e ? atoi(e) : 8192; }
    if ((long)H * W >= tiled_min) {
        dim3 grid(mv_ceil_div(W, LC_T), mv_ceil_div(H, LC_T), B * 3);
        hipLaunchKernelGGL(local_corr81_kernel, grid, dim3(256), 0, s, first, second, out, C, H, W);
        return mv_launch_status();
    }
    // channel split: enough threads to fill the chip (>= 128 k), at least 8 channels per thread
    int S = 1;
    while (S < 16 && total * S < 131072 && C / (2 * S) >= 8) S *= 2;
#define MV_LC_DIRECT(SS)                                                                                               \
    hipLaunchKernelGGL(local_corr81_direct<SS>, dim3((unsigned)((total * SS + 255) / 256)), dim3(256), 0, s, first, second, \
                       out, C, H, W, total)
    switch (S) {
        case 1: MV_LC_DIRECT(1); break;
        case 2: MV_LC_DIRECT(2); break;
        case 4: MV_LC_DIRECT(4); break;
        case 8: MV_LC_DIRECT(8); break;
        default: MV_LC_DIRECT(16); break;
    }
#undef MV_LC_DIRECT
    return mv_launch_status();
}
