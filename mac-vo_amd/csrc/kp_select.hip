// A10 / A11 / MappingPointSelector — dense half of the covariance-aware keypoint selectors
//
// Replaces Module/KeypointSelector.py:
//   CovAwareSelector_NoDepth.select_point :362-400, CovAwareSelector.select_point :260-327,
//   MappingPointSelector.select_point :87-97
// i.e. quality map -> k x k min-NMS (max_pool2d(-q) semantics: -inf padding, NaN propagates, equality test)
// -> border mask -> (nan)median * 1.5 thresholds (strict '<', fp32 compare against the fp32-rounded scalar)
// -> optional validity masks -> torch.nonzero row-major candidate order.
// The host keeps torch.randperm (global CPU generator) so selected indices are bit-exact.
//
// gfx950 design (HBM-bound at ~4-9 MB/frame, in practice launch/latency-bound):
//   kernel 1 (grid of 64x16 tiles): quality tile + halo in LDS, NMS, one 64-bit __ballot word per
//            64-pixel row segment (coalesced 8-B stores) for `nms` and for the threshold-independent
//            part of the mask; the median populations are appended with one wave-aggregated atomic.
//   kernel 2 (one 1024-thread workgroup): 4-pass 8-bit radix select (LDS histograms) for the lower
//            median(s), thresholds, then an ordered stream compaction of the bit words
//            (per-thread popcount -> workgroup exclusive scan -> in-order writes).
// No host synchronisation inside; the caller reads back out_count when it needs n for randperm.
#include "common.h"
#include <math.h>

namespace {

constexpr int TILE_W = 64;
constexpr int TILE_H = 16;
constexpr int MAX_R = 7;

struct KpWs {
    unsigned long long* nms_bits;
    unsigned long long* cand_bits;
    float* pop_a;   // flow-quality population  (q[nms])
    float* pop_b;   // depth0_cov population    (depth0_cov[nms])
    int* counters;  // [0] = population size
};

__device__ __forceinline__ float flow_quality(const float* __restrict__ fc, int plane, int idx) {
    // (c0 + c1) - 2*c2, each op rounded to fp32 as torch does (2*c2 is exact)
    const float c0 = fc[idx], c1 = fc[plane + idx], c2 = fc[2 * plane + idx];
    return (c0 + c1) - 2.f * c2;
}

__device__ __forceinline__ float quality_at(int mode, const float* __restrict__ fc, const float* __restrict__ d0c,
                                            const float* __restrict__ d1c, int plane, int idx) {
    if (mode == MV_KP_NODEPTH) return flow_quality(fc, plane, idx);
    float q = d0c[idx] + d1c[idx];
    if (fc) q = q * flow_quality(fc, plane, idx);
    return q;
}

__global__ __launch_bounds__(256) void kp_nms_kernel(const float* __restrict__ fc, const float* __restrict__ d0,
                                                      const float* __restrict__ d0c, const float* __restrict__ d1,
                                                      const float* __restrict__ d1c,
                                                      const uint8_t* __restrict__ mask_a,
                                                      const uint8_t* __restrict__ mask_b, mvKpSelectParams p,
                                                      KpWs ws, int words_per_row) {
    __shared__ float tile[TILE_H + 2 * MAX_R][TILE_W + 2 * MAX_R + 1];
    const int H = p.H, W = p.W, plane = H * W;
    const int r = p.kernel_size >> 1;
    const int x0 = blockIdx.x * TILE_W, y0 = blockIdx.y * TILE_H;
    const int tx = threadIdx.x, ty = threadIdx.y;  // tx = lane (0..63), ty = wave (0..3)
    const int tid = ty * 64 + tx;
    const bool mapping = p.mode == MV_KP_MAPPING;

    if (!mapping) {
        const int tw = TILE_W + 2 * r, th = TILE_H + 2 * r;
        for (int e = tid; e < tw * th; e += 256) {
            const int ly = e / tw, lx = e - ly * tw;
            const int gx = x0 + lx - r, gy = y0 + ly - r;
            float q = INFINITY;  // out-of-image == -inf padding of max_pool2d(-q): never the minimum
            if (gx >= 0 && gx < W && gy >= 0 && gy < H) q = quality_at(p.mode, fc, d0c, d1c, plane, gy * W + gx);
            tile[ly][lx] = q;
        }
        __syncthreads();
    }

#pragma unroll
    for (int it = 0; it < TILE_H / 4; ++it) {
        const int ly = ty * (TILE_H / 4) + it;
        const int gx = x0 + tx, gy = y0 + ly;
        const bool inimg = gx < W && gy < H;
        const int idx = gy * W + gx;
        bool nms = false, cand = false;
        float q = 0.f;
        if (inimg) {
            const bool border = p.mask_width > 0 && gx >= p.mask_width && gx < W - p.mask_width &&
                                gy >= p.mask_width && gy < H - p.mask_width;
            if (mapping) {
                cand = border && (d0[idx] < p.max_depth) && (d0c[idx] < p.max_depth_cov);
            } else {
                q = tile[ly + r][tx + r];
                float m = INFINITY;
                bool has_nan = false;
                for (int dy = 0; dy <= 2 * r; ++dy)
                    for (int dx = 0; dx <= 2 * r; ++dx) {
                        const float v = tile[ly + dy][tx + dx];
                        has_nan |= (v != v);
                        m = fminf(m, v);
                    }
                nms = !has_nan && (q == m);  // q NaN => has_nan
                cand = nms && border;
                if (cand && p.mode == MV_KP_FULL) cand = (d0[idx] < p.max_depth) && (d1[idx] < p.max_depth);
            }
            if (cand && mask_a) cand = mask_a[idx] != 0;
            if (cand && mask_b) cand = mask_b[idx] != 0;
        }
        const unsigned long long nms_word = __ballot(nms);
        const unsigned long long cand_word = __ballot(cand);
        if (tx == 0 && gy < H) {
            ws.nms_bits[(size_t)gy * words_per_row + blockIdx.x] = nms_word;
            ws.cand_bits[(size_t)gy * words_per_row + blockIdx.x] = cand_word;
        }
        if (!mapping && nms_word) {
            // wave-aggregated append of the median population(s); order is irrelevant for a median
            const int cnt = __popcll(nms_word);
            int base = 0;
            if (tx == 0) base = atomicAdd(&ws.counters[0], cnt);
            base = __shfl(base, 0, 64);
            if (nms) {
                const int rank = __popcll(nms_word & ((1ull << tx) - 1ull));
                if (p.mode == MV_KP_NODEPTH) {
                    ws.pop_a[base + rank] = q;
                } else {
                    if (fc) ws.pop_a[base + rank] = flow_quality(fc, plane, idx);
                    ws.pop_b[base + rank] = d0c[idx];
                }
            }
        }
    }
}

// order-preserving float -> uint key; every NaN sorts last
__device__ __forceinline__ unsigned float_key(float f) {
    if (f != f) return 0xFFFFFFFFu;
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(unsigned k) {
    const unsigned u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}

// Lower median ((m-1)/2-th smallest of the m non-NaN values) == torch.median / torch.nanmedian.
// Whole workgroup participates; returns NaN when there is no non-NaN value.
__device__ float block_nanmedian(const float* __restrict__ vals, int n, unsigned* hist /*[256]*/, int* sh /*[4]*/) {
    const int tid = threadIdx.x, nt = blockDim.x;
    // count NaNs
    if (tid == 0) sh[0] = 0;
    __syncthreads();
    int local_nan = 0;
    for (int i = tid; i < n; i += nt) local_nan += (vals[i] != vals[i]);
    local_nan = wave_sum(local_nan);
    if ((tid & 63) == 0 && local_nan) atomicAdd(&sh[0], local_nan);
    __syncthreads();
    const int m = n - sh[0];
    __syncthreads();
    if (m <= 0) return NAN;
    int k = (m - 1) >> 1;
    unsigned prefix = 0, mask = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = tid; i < 256; i += nt) hist[i] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += nt) {
            const unsigned key = float_key(vals[i]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int acc = 0, bkt = 0;
            for (; bkt < 256; ++bkt) {
                const int c = (int)hist[bkt];
                if (acc + c > k) break;
                acc += c;
            }
            sh[1] = bkt;
            sh[2] = k - acc;
        }
        __syncthreads();
        prefix |= ((unsigned)sh[1]) << shift;
        mask |= 255u << shift;
        k = sh[2];
        __syncthreads();
    }
    return key_float(prefix);
}

__global__ __launch_bounds__(1024) void kp_compact_kernel(const float* __restrict__ fc,
                                                           const float* __restrict__ d0c, mvKpSelectParams p,
                                                           KpWs ws, int words_per_row, int32_t* __restrict__ out_cand,
                                                           int32_t* __restrict__ out_count,
                                                           float* __restrict__ out_stats) {
    __shared__ unsigned hist[256];
    __shared__ int sh[4];
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x;
    const int H = p.H, W = p.W, plane = H * W;
    const int n_pop = (p.mode == MV_KP_MAPPING) ? 0 : ws.counters[0];

    float med_f = NAN, thr_f = INFINITY, med_d = NAN, thr_d = INFINITY;
    const bool use_f = (p.mode == MV_KP_NODEPTH) || (p.mode == MV_KP_FULL && fc != nullptr);
    const bool use_d = (p.mode == MV_KP_FULL);
    if (use_f) {
        med_f = block_nanmedian(ws.pop_a, n_pop, hist, sh);
        // python: min(max_match_cov, median * 1.5) in double, then the fp32 compare rounds it to fp32:
        // == fp32 min of fp32-rounded operands (rounding is monotonic; med*1.5 is exact in double).
        const float prod = med_f * 1.5f;
        thr_f = (prod < p.max_match_cov) ? prod : p.max_match_cov;  // python min(a, b): b if b < a else a
    }
    if (use_d) {
        med_d = block_nanmedian(ws.pop_b, n_pop, hist, sh);
        const float prod = med_d * 1.5f;
        thr_d = (prod < p.max_depth_cov) ? prod : p.max_depth_cov;
    }

    // ---- ordered compaction of the candidate words
    const int n_words = H * words_per_row;
    const int per = (n_words + 1023) / 1024;
    const int w_begin = tid * per, w_end = min(w_begin + per, n_words);

    auto survives = [&](int idx) -> bool {
        bool ok = true;
        if (use_f) ok = flow_quality(fc, plane, idx) < thr_f;
        if (ok && use_d) ok = d0c[idx] < thr_d;
        return ok;
    };

    int cnt = 0;
    for (int w = w_begin; w < w_end; ++w) {
        unsigned long long bits = ws.cand_bits[w];
        const int row = w / words_per_row, col0 = (w - row * words_per_row) * 64;
        while (bits) {
            const int bpos = __ffsll((long long)bits) - 1;
            bits &= bits - 1;
            if (p.mode == MV_KP_MAPPING || survives(row * W + col0 + bpos)) ++cnt;
        }
    }
    // workgroup exclusive scan of cnt
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if ((tid & 63) >= o) incl += v;
    }
    if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
    __syncthreads();
    int wave_off = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < (tid >> 6)) wave_off += wave_tot[w];
        total += wave_tot[w];
    }
    int pos = wave_off + incl - cnt;
    for (int w = w_begin; w < w_end; ++w) {
        unsigned long long bits = ws.cand_bits[w];
        const int row = w / words_per_row, col0 = (w - row * words_per_row) * 64;
        while (bits) {
            const int bpos = __ffsll((long long)bits) - 1;
            bits &= bits - 1;
            const int idx = row * W + col0 + bpos;
            if (p.mode == MV_KP_MAPPING || survives(idx)) out_cand[pos++] = idx;
        }
    }
    if (tid == 0) {
        out_count[0] = total;
        out_count[1] = n_pop;
        out_count[2] = 0;
        out_count[3] = 0;
        out_stats[0] = med_f;
        out_stats[1] = thr_f;
        out_stats[2] = med_d;
        out_stats[3] = thr_d;
    }
}

__global__ void kp_gather_kernel(const int32_t* __restrict__ cand, const int64_t* __restrict__ perm, int n_sel,
                                 int W, int64_t* __restrict__ out_uv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_sel) {
        const int lin = cand[perm[i]];
        out_uv[2 * i + 0] = lin % W;  // u
        out_uv[2 * i + 1] = lin / W;  // v
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

extern "C" size_t mv_kp_select_workspace_bytes(int H, int W) {
    if (H <= 0 || W <= 0) return 0;
    const size_t words = (size_t)H * mv_ceil_div(W, 64);
    return 2 * align_up(words * 8, 256) + 2 * align_up((size_t)H * W * 4, 256) + 256;
}

extern "C" int mv_kp_select(const float* flow_cov, const float* depth0, const float* depth0_cov,
                            const float* depth1, const float* depth1_cov, const uint8_t* mask_a,
                            const uint8_t* mask_b, const mvKpSelectParams* params, void* workspace,
                            size_t workspace_bytes, int32_t* out_cand, int32_t* out_count, float* out_stats,
                            mvStream_t stream) {
    MV_CHECK_ARG(params && workspace && out_cand && out_count && out_stats);
    const mvKpSelectParams p = *params;
    MV_CHECK_ARG(p.H > 0 && p.W > 0 && p.mask_width >= 0);
    if (p.mode == MV_KP_NODEPTH) {
        MV_CHECK_ARG(flow_cov);
    } else if (p.mode == MV_KP_FULL) {
        MV_CHECK_ARG(depth0 && depth0_cov && depth1 && depth1_cov);
    } else if (p.mode == MV_KP_MAPPING) {
        MV_CHECK_ARG(depth0 && depth0_cov);
    } else {
        return MV_ERR_INVALID_ARG;
    }
    if (p.mode != MV_KP_MAPPING) {
        MV_CHECK_ARG(p.kernel_size >= 1 && (p.kernel_size & 1));
        if (p.kernel_size > 2 * MAX_R + 1) return MV_ERR_UNSUPPORTED;
    }
    if (workspace_bytes < mv_kp_select_workspace_bytes(p.H, p.W)) return MV_ERR_WORKSPACE;
    if (((uintptr_t)workspace & 7) != 0) return MV_ERR_INVALID_ARG;

    const int wpr = mv_ceil_div(p.W, 64);
    const size_t words = (size_t)p.H * wpr;
    char* base = (char*)workspace;
    KpWs ws;
    ws.nms_bits = (unsigned long long*)base;
    base += align_up(words * 8, 256);
    ws.cand_bits = (unsigned long long*)base;
    base += align_up(words * 8, 256);
    ws.pop_a = (float*)base;
    base += align_up((size_t)p.H * p.W * 4, 256);
    ws.pop_b = (float*)base;
    base += align_up((size_t)p.H * p.W * 4, 256);
    ws.counters = (int*)base;

    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(ws.counters, 0, 16, s) != hipSuccess) return MV_ERR_LAUNCH;
    dim3 grid(wpr, mv_ceil_div(p.H, TILE_H)), block(64, 4);
    hipLaunchKernelGGL(kp_nms_kernel, grid, block, 0, s, flow_cov, depth0, depth0_cov, depth1, depth1_cov, mask_a,
                       mask_b, p, ws, wpr);
    hipLaunchKernelGGL(kp_compact_kernel, dim3(1), dim3(1024), 0, s, flow_cov, depth0_cov, p, ws, wpr, out_cand,
                       out_count, out_stats);
    return mv_launch_status();
}

extern "C" int mv_kp_gather(const int32_t* cand, const int64_t* perm, int n_sel, int W, int64_t* out_uv,
                            mvStream_t stream) {
    MV_CHECK_ARG(n_sel >= 0 && W > 0);
    if (n_sel == 0) return MV_OK;
    MV_CHECK_ARG(cand && perm && out_uv);
    hipLaunchKernelGGL(kp_gather_kernel, dim3(mv_ceil_div(n_sel, 256)), dim3(256), 0, (hipStream_t)stream, cand, perm,
                       n_sel, W, out_uv);
    return mv_launch_status();
}
