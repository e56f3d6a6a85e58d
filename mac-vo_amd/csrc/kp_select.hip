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
// gfx950 design (4-9 MB of HBM traffic per frame; in practice launch/latency-bound, so: few launches, no
// sparse gathers on the single-workgroup stage, one atomic per workgroup)
//   kernel 1 (grid of 64x16 tiles, 256 threads): quality tile + halo in LDS (all global loads of a thread in flight
//            before the first LDS store), SEPARABLE NaN-propagating min (k + k LDS reads per pixel instead of k*k),
//            equality test, one 64-bit __ballot word per 64-pixel row segment for the threshold-independent part of
//            the mask.  Every NMS pixel also emits a 12-byte record {bit address | candidate flag, flow quality,
//            depth0 variance}; a workgroup reserves its record range with ONE global atomic (LDS-aggregated).  That
//            atomic is the kernel's longest dependency (measured: ~6 of its 13 us are 300 workgroups waiting for the
//            return value of a same-address device-scope atomic); removing it needs per-workgroup record regions and a
//            re-indexing pass in kernel 2 that costs about half of what it saves — left as is.
//   kernel 2 (one 1024-thread workgroup; 256 / 512 threads measured 3x / 1.4x slower): every global read is issued
//            up front (records -> registers, candidate words -> registers -> dynamic LDS); lower (nan)median(s) by
//            BUCKET REFINEMENT on order-preserving keys (min/max -> 2048 linear buckets over the population's own range
//            -> direct ranking once the bucket holds <= 256 keys: one histogram pass in the common case, no hot bin);
//            records failing a threshold clear their bit with an LDS atomicAnd; ordered stream compaction of the bit
//            words (popcount -> workgroup scan -> in-order writes); the record counter is left zeroed for the next
//            call.  23.5 us (3-pass radix select re-reading global memory) -> 11.7 us.
//   MAPPING mode (no medians, ~10^5 candidates): kernel 1, a one-workgroup popcount scan, and a wave-per-word emit.
// No host synchronisation inside; the caller reads back out_count when it needs n for randperm.
#include "common.h"
#include <math.h>

namespace {

#ifdef MV_KP_PROFILE
__device__ long long g_kp_stamps[16];
#define KP_STAMP(i) do { if (threadIdx.x == 0) g_kp_stamps[i] = wall_clock64(); } while (0)
#else
#define KP_STAMP(i)
#endif

constexpr int TILE_W = 64;
constexpr int TILE_H = 16;
constexpr int MAX_R = 7;
constexpr unsigned CAND_FLAG = 0x80000000u;

struct KpWs {
    unsigned long long* cand_bits;
    unsigned* rec_idx;  // (candidate-word index << 6 | bit) | CAND_FLAG: where this NMS pixel's bit lives in cand_bits
    float* rec_q;       // flow quality of the NMS pixel      (population of the flow-cov median)
    float* rec_d;       // depth0 variance of the NMS pixel   (population of the depth-cov median, FULL only)
    int* counters;      // [0] = number of records (= NMS pixels)
    size_t lane_bytes;  // lane-batched launches: lane l works in the workspace copy at + l * lane_bytes

    __device__ __forceinline__ KpWs lane(int l) const {
        const size_t o = (size_t)l * lane_bytes;
        return KpWs{(unsigned long long*)((char*)cand_bits + o), (unsigned*)((char*)rec_idx + o), (float*)((char*)rec_q + o),
                    (float*)((char*)rec_d + o), (int*)((char*)counters + o), lane_bytes};
    }
};

__device__ __forceinline__ float flow_quality(const float* __restrict__ fc, int plane, int idx) {
    // (c0 + c1) - 2*c2, each op rounded to fp32 as torch does (2*c2 is exact)
    const float c0 = fc[idx], c1 = fc[plane + idx], c2 = fc[2 * plane + idx];
    return (c0 + c1) - 2.f * c2;
}

// min that propagates NaN from either side (max_pool2d keeps a NaN once seen)
__device__ __forceinline__ float nanmin(float a, float b) { return (a < b || a != a) ? a : b; }

// FUSE (MV_KP_NODEPTH only): the frontend epilogue rides along — the quality q = sigma_uu + sigma_vv of the tile and its halo is
// computed from the network's log-sigma planes (the same expf as the epilogue's, so the same bits as reading match_cov back),
// and every thread writes the epilogue outputs of its own four pixels: one launch and one pass over the maps less per frame.
template <bool FUSE>
__global__ __launch_bounds__(256) void kp_nms_kernel(const float* __restrict__ fc, const float* __restrict__ d0,
                                                      const float* __restrict__ d0c, const float* __restrict__ d1,
                                                      const float* __restrict__ d1c,
                                                      const uint8_t* __restrict__ mask_a,
                                                      const uint8_t* __restrict__ mask_b, mvKpSelectParams p,
                                                      KpWs ws_all, int words_per_row, mvEpiArgs ef) {
    MV_CHAIN_KERNEL_PRIO();
    // lane-batched: blockIdx.z = lane (independent frame); maps are [lanes, ch, H, W], one workspace copy per lane
    const KpWs ws = ws_all.lane(blockIdx.z);
    {
        const size_t lp = (size_t)blockIdx.z * p.H * p.W;
        if (fc) fc += 3 * lp;
        if (d0) d0 += lp;
        if (d0c) d0c += lp;
        if (d1) d1 += lp;
        if (d1c) d1c += lp;
        if (mask_a) mask_a += lp;
        if (mask_b) mask_b += lp;
        if (FUSE) {   // inputs [lanes, 2, 2, H, W], outputs [lanes, ch, H, W]
            ef.flow += 4 * lp; ef.logcov += 4 * lp;
            if (ef.disparity) ef.disparity += lp;
            if (ef.disparity_cov) ef.disparity_cov += lp;
            if (ef.depth) ef.depth += lp;
            if (ef.depth_cov) ef.depth_cov += lp;
            if (ef.bad_mask) ef.bad_mask += lp;
            if (ef.match_flow) ef.match_flow += 2 * lp;
            if (ef.match_cov) ef.match_cov += 3 * lp;
        }
    }
    __shared__ float tile[TILE_H + 2 * MAX_R][TILE_W + 2 * MAX_R + 1];
    __shared__ float hmin[TILE_H + 2 * MAX_R][TILE_W + 1];
    __shared__ int wg_count, wg_base;
    const int H = p.H, W = p.W, plane = H * W;
    const int r = p.kernel_size >> 1;
    const int x0 = blockIdx.x * TILE_W, y0 = blockIdx.y * TILE_H;
    const int tx = threadIdx.x, ty = threadIdx.y;  // tx = lane (0..63), ty = wave (0..3)
    const int tid = ty * 64 + tx;
    const bool mapping = p.mode == MV_KP_MAPPING;
    if (tid == 0) wg_count = 0;

    if (!mapping) {
        const int tw = TILE_W + 2 * r, th = TILE_H + 2 * r;
        // stage the quality tile + halo: all global loads of a thread are issued before the first LDS store
        constexpr int NST = ((TILE_W + 2 * MAX_R) * (TILE_H + 2 * MAX_R) + 255) / 256;   // 9 for the 78 x 30 maximum
        float qv[NST];
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int e = i * 256 + tid;
            const int ly = e / tw, lx = e - ly * tw;
            const int gx = x0 + lx - r, gy = y0 + ly - r;
            float q = INFINITY;  // out-of-image == -inf padding of max_pool2d(-q): never the minimum
            if (e < tw * th && gx >= 0 && gx < W && gy >= 0 && gy < H) {
                const int idx = gy * W + gx;
                if (FUSE) {
                    // flow_quality of (c0, c1, 0): (c0 + c1) - 2 * 0 = c0 + c1 exactly
                    const float l0 = ef.logcov[2 * plane + idx], l1 = ef.logcov[3 * plane + idx];
                    const float c0 = ef.cov_is_log ? expf(l0 * 2.f) : l0, c1 = ef.cov_is_log ? expf(l1 * 2.f) : l1;
                    q = (c0 + c1) - 2.f * 0.f;
                } else if (p.mode == MV_KP_NODEPTH) {
                    q = flow_quality(fc, plane, idx);
                } else {
                    q = d0c[idx] + d1c[idx];
                    if (fc) q = q * flow_quality(fc, plane, idx);
                }
            }
            qv[i] = q;
        }
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int e = i * 256 + tid;
            const int ly = e / tw, lx = e - ly * tw;
            if (e < tw * th) tile[ly][lx] = qv[i];
        }
        __syncthreads();
        // horizontal pass: hmin[ly][x] = nanmin over tile[ly][x .. x+2r]
        for (int e = tid; e < th * TILE_W; e += 256) {
            const int ly = e >> 6, lx = e & 63;
            float m = tile[ly][lx];
            for (int dx = 1; dx <= 2 * r; ++dx) m = nanmin(m, tile[ly][lx + dx]);
            hmin[ly][lx] = m;
        }
    }
    __syncthreads();

    bool nms_px[TILE_H / 4];
    bool cand_px[TILE_H / 4];
    float q_px[TILE_H / 4];
    int my_nms = 0;
#pragma unroll
    for (int it = 0; it < TILE_H / 4; ++it) {
        const int ly = ty * (TILE_H / 4) + it;
        const int gx = x0 + tx, gy = y0 + ly;
        const bool inimg = gx < W && gy < H;
        const int idx = gy * W + gx;
        bool nms = false, cand = false;
        float q = 0.f;
        if (FUSE && inimg) mv_epilogue_pixel(ef, plane, idx);
        if (inimg) {
            const bool border = p.mask_width > 0 && gx >= p.mask_width && gx < W - p.mask_width &&
                                gy >= p.mask_width && gy < H - p.mask_width;
            if (mapping) {
                cand = border && (d0[idx] < p.max_depth) && (d0c[idx] < p.max_depth_cov);
            } else {
                q = tile[ly + r][tx + r];
                float m = hmin[ly][tx];
                for (int dy = 1; dy <= 2 * r; ++dy) m = nanmin(m, hmin[ly + dy][tx]);
                nms = (q == m);  // false whenever the window holds a NaN (m is NaN then) or q itself is NaN
                cand = nms && border;
                if (cand && p.mode == MV_KP_FULL) cand = (d0[idx] < p.max_depth) && (d1[idx] < p.max_depth);
            }
            if (cand && mask_a) cand = mask_a[idx] != 0;
            if (cand && mask_b) cand = mask_b[idx] != 0;
        }
        const unsigned long long cand_word = __ballot(cand);
        if (tx == 0 && gy < H) ws.cand_bits[(size_t)gy * words_per_row + blockIdx.x] = cand_word;
        nms_px[it] = nms;
        cand_px[it] = cand;
        q_px[it] = q;
        my_nms += nms;
    }
    if (mapping) return;

    // ---- records: LDS-aggregated reservation, one global atomic per workgroup
    int my_off = 0;
    if (my_nms) my_off = atomicAdd(&wg_count, my_nms);
    __syncthreads();
#ifdef MV_KP_FAKE_ATOMIC
    if (tid == 0) wg_base = (blockIdx.y * gridDim.x + blockIdx.x) * 16;   // timing experiment only (wrong results)
#else
    if (tid == 0) wg_base = wg_count ? atomicAdd(&ws.counters[0], wg_count) : 0;
#endif
    __syncthreads();
    int pos = wg_base + my_off;
#pragma unroll
    for (int it = 0; it < TILE_H / 4; ++it) {
        if (nms_px[it]) {
            const int gx = x0 + tx, gy = y0 + ty * (TILE_H / 4) + it;
            const int idx = gy * W + gx;
            ws.rec_idx[pos] = (unsigned)(((gy * words_per_row + blockIdx.x) << 6) | tx) | (cand_px[it] ? CAND_FLAG : 0u);
            if (p.mode == MV_KP_NODEPTH) {
                ws.rec_q[pos] = q_px[it];
            } else {
                ws.rec_q[pos] = fc ? flow_quality(fc, plane, idx) : 0.f;
                ws.rec_d[pos] = d0c[idx];
            }
            ++pos;
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

#ifndef MV_KP_FINISH_THREADS
#define MV_KP_FINISH_THREADS 1024
#endif

constexpr int RANK_CAP = 256;                        // bucket size at which selection switches to direct ranking
constexpr unsigned NAN_KEY = 0xFFFFFFFFu;

// workgroup exclusive scan of one int per thread; returns the exclusive prefix and the total
template <int NT>
__device__ __forceinline__ int block_exclusive_scan(int v, int* wave_tot /*[NT/64] LDS*/, int& total) {
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if ((tid & 63) >= o) incl += t;
    }
    __syncthreads();  // protect wave_tot reuse
    if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
    __syncthreads();
    int off = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int t = wave_tot[w];
        if (w < (tid >> 6)) off += t;
        total += t;
    }
    return off + incl - v;
}

template <int NT, bool CACHED, int RPT, typename F>
__device__ __forceinline__ void for_each_key(const float* __restrict__ vals, int n, const unsigned (&keys)[RPT], F&& f) {
    const int tid = threadIdx.x;
    if (CACHED) {
        const int nr = (n + NT - 1) / NT;   // register slots that hold data (uniform)
#pragma unroll
        for (int r = 0; r < RPT; ++r)
            if (r < nr && r * NT + tid < n) f(keys[r]);
    } else {
        for (int i = tid; i < n; i += NT) f(float_key(vals[i]));
    }
}

struct MedianLds {
    unsigned hist[2048];
    unsigned lst[RANK_CAP];
    unsigned red_lo[16], red_hi[16];
    int red_cnt[16];
    int wave_tot[16];
    int sh[4];
    unsigned res;
};

// Lower median ((m-1)/2-th smallest of the m non-NaN values) == torch.median / torch.nanmedian of the population, by
// bucket refinement on the order-preserving keys: [lo, hi] starts as the population's own range (so the 2048 linear
// buckets are spread over the values that exist: no hot histogram bin, unlike a fixed sign/exponent digit), the bucket
// holding rank k becomes the next range, and as soon as that bucket holds <= 256 keys they are ranked directly.
// Typical cost: one min/max reduction, one histogram pass, one scan, one tiny ranking step.  The population is read
// from registers (CACHED: thread t holds keys t, t + NT, ...) or re-read from global memory (any size).
template <int NT, bool CACHED, int RPT>
__device__ float block_nanmedian(const float* __restrict__ vals, int n, const unsigned (&keys)[RPT], MedianLds& L) {
    constexpr int NW = NT / 64, BPT = 2048 / NT;
    const int tid = threadIdx.x;
    unsigned lo = NAN_KEY, hi = 0u;
    int cnt = 0;
    for_each_key<NT, CACHED, RPT>(vals, n, keys, [&](unsigned k) {
        if (k != NAN_KEY) {
            lo = min(lo, k);
            hi = max(hi, k);
            ++cnt;
        }
    });
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, (unsigned)__shfl_xor((int)lo, o, 64));
        hi = max(hi, (unsigned)__shfl_xor((int)hi, o, 64));
        cnt += __shfl_xor(cnt, o, 64);
    }
    __syncthreads();   // L may still be in use by a previous call
    if ((tid & 63) == 0) {
        L.red_lo[tid >> 6] = lo;
        L.red_hi[tid >> 6] = hi;
        L.red_cnt[tid >> 6] = cnt;
    }
    __syncthreads();
    cnt = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        lo = min(lo, L.red_lo[w]);
        hi = max(hi, L.red_hi[w]);
        cnt += L.red_cnt[w];
    }
    if (cnt <= 0) return NAN;
    int k = (cnt - 1) >> 1;
    for (;;) {   // every quantity below is workgroup-uniform
        if (lo == hi) return key_float(lo);
        const unsigned span = hi - lo;
        const int s = max(0, 32 - __clz((int)span) - 11);   // (span >> s) < 2048
#pragma unroll
        for (int j = 0; j < BPT; ++j) L.hist[j * NT + tid] = 0;
        if (tid == 0) L.sh[3] = 0;
        __syncthreads();
        for_each_key<NT, CACHED, RPT>(vals, n, keys, [&](unsigned kk) {
            if (kk >= lo && kk <= hi) atomicAdd(&L.hist[(kk - lo) >> s], 1u);
        });
        __syncthreads();
        int c[BPT], sum = 0;
#pragma unroll
        for (int j = 0; j < BPT; ++j) {
            c[j] = (int)L.hist[tid * BPT + j];
            sum += c[j];
        }
        int total;
        const int excl = block_exclusive_scan<NT>(sum, L.wave_tot, total);
        if (excl <= k && k < excl + sum) {  // exactly one thread owns the bucket of rank k
            int before = excl;
#pragma unroll
            for (int j = 0; j < BPT; ++j) {
                if (k >= before && k < before + c[j]) {
                    L.sh[0] = tid * BPT + j;
                    L.sh[1] = c[j];
                    L.sh[2] = k - before;
                }
                before += c[j];
            }
        }
        __syncthreads();
        const unsigned b = (unsigned)L.sh[0];
        const int cb = L.sh[1];
        k = L.sh[2];
        const unsigned nlo = lo + (b << s);
        const unsigned w = s ? ((1u << s) - 1u) : 0u;
        const unsigned nhi = (hi - nlo < w) ? hi : nlo + w;
        if (cb <= RANK_CAP) {
            for_each_key<NT, CACHED, RPT>(vals, n, keys, [&](unsigned kk) {
                if (kk >= nlo && kk <= nhi) L.lst[atomicAdd(&L.sh[3], 1)] = kk;
            });
            __syncthreads();
            for (int i = tid; i < cb; i += NT) {
                const unsigned e = L.lst[i];
                int rank = 0;
                for (int j = 0; j < cb; ++j) {
                    const unsigned o = L.lst[j];
                    rank += (o < e) || (o == e && j < i);
                }
                if (rank == k) L.res = e;
            }
            __syncthreads();
            return key_float(L.res);
        }
        lo = nlo;
        hi = nhi;
        __syncthreads();   // sh / hist are rewritten by the next round
    }
}

// One workgroup of NT threads.  FAST (n_rec <= RPT * NT and H * words_per_row <= WPT * NT): every global
// array is read exactly once with all loads in flight (records -> registers, candidate words -> registers -> LDS), the
// threshold pass clears bits with LDS atomics and the compaction reads LDS.  Otherwise the same steps run against
// global memory.  Few fat threads on purpose: the work is a few thousand elements on ONE compute unit, where every
// wave-level instruction costs 4 cycles and per-wave fixed costs (scans, loop control) multiply with the wave count.
template <int NT, bool FAST, int RPT, int WPT, bool USE_D>
__device__ __forceinline__ void kp_finish_body(const mvKpSelectParams& p, const KpWs& ws, int has_flow, int words_per_row,
                                               int32_t* __restrict__ out_cand, int32_t* __restrict__ out_count,
                                               float* __restrict__ out_stats, int n_rec, MedianLds& L,
                                               unsigned long long* lds_words) {
    constexpr int RD = USE_D ? RPT : 1;   // the depth-variance keys only exist in FULL mode
    const int tid = threadIdx.x;
    const int H = p.H, W = p.W;
    const bool mapping = p.mode == MV_KP_MAPPING;
    const bool use_f = (p.mode == MV_KP_NODEPTH) || (p.mode == MV_KP_FULL && has_flow);
    const bool use_d = USE_D && (p.mode == MV_KP_FULL);
    const int n_words = H * words_per_row;

    // FAST: every global read of this kernel is issued here, before anything waits: the record arrays are read for all
    // register slots without knowing n_rec yet (they are plane-sized, so the reads stay inside the workspace), the
    // candidate words go to registers and are parked in LDS only after the medians
    unsigned kq[RPT], kd[RD], ridx[RPT];
    unsigned long long wreg[WPT];
    KP_STAMP(0);
    if (FAST) {
        const int cap = H * W;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int i = min(r * NT + tid, cap - 1);
            ridx[r] = mapping ? 0u : ws.rec_idx[i];
            kq[r] = use_f ? __float_as_uint(ws.rec_q[i]) : 0u;
            if (USE_D) kd[r] = use_d ? __float_as_uint(ws.rec_d[i]) : 0u;
        }
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const int w = j * NT + tid;
            wreg[j] = w < n_words ? ws.cand_bits[w] : 0ull;
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const bool live = r * NT + tid < n_rec;
            ridx[r] = live ? ridx[r] : 0u;
            kq[r] = live ? float_key(__uint_as_float(kq[r])) : NAN_KEY;
            if (USE_D) kd[r] = live ? float_key(__uint_as_float(kd[r])) : NAN_KEY;
        }
    }

    float med_f = NAN, thr_f = INFINITY, med_d = NAN, thr_d = INFINITY;
    KP_STAMP(1);
    if (use_f) {
        med_f = block_nanmedian<NT, FAST, RPT>(ws.rec_q, n_rec, kq, L);
        // python: min(max_match_cov, median * 1.5) in double, then the fp32 compare rounds it to fp32:
        // == fp32 min of fp32-rounded operands (rounding is monotonic; med*1.5 is exact in double).
        const float prod = med_f * 1.5f;
        thr_f = (prod < p.max_match_cov) ? prod : p.max_match_cov;  // python min(a, b): b if b < a else a
    }
    if (use_d) {
        med_d = block_nanmedian<NT, FAST, RD>(ws.rec_d, n_rec, kd, L);
        const float prod = med_d * 1.5f;
        thr_d = (prod < p.max_depth_cov) ? prod : p.max_depth_cov;
    }

    KP_STAMP(2);
    const int per = (n_words + NT - 1) / NT;   // thread t owns the contiguous words [t * per, (t + 1) * per) in the compaction
    unsigned long long cw[WPT];                // FAST: those words, after the clears
    if (FAST) {
        // The words pass through LDS in NPASS chunks (park -> atomic clears -> read back contiguous): the kernel then needs
        // <= 20 KB of LDS beside its 9 KB of median scratch instead of 40-120 KB.  That matters because this single workgroup
        // is launched while the NEXT frame's volume GEMM owns the chip (four 32-KB workgroups per CU = 128 of 160 KB): with a
        // 50-KB footprint it could not be placed until that GEMM's grid had drained, and the candidate count — which the host
        // needs before it can enqueue the frame after — arrived ~200 us late (unprofiled timeline).
        constexpr int NPASS = WPT > 5 * 1024 / NT ? 8 : 2;   // 2 chunks for 640x480-class images, 8 for the large variant
        constexpr int TPP = NT / NPASS;        // threads whose contiguous ranges make up one chunk
        unsigned fail = 0;                     // bit r: candidate record in slot r fails a threshold
        if (!mapping) {
            const int nr = (n_rec + NT - 1) / NT;
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                if (r < nr && (ridx[r] & CAND_FLAG)) {
                    // cached keys are mapped back to the exact floats (a NaN comes back as a NaN): fp32 `<` as on the originals
                    bool ok = true;
                    if (use_f) ok = key_float(kq[r]) < thr_f;
                    if (USE_D) { if (ok && use_d) ok = key_float(kd[USE_D ? r : 0]) < thr_d; }
                    if (!ok) fail |= 1u << r;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) cw[i] = 0ull;
#pragma unroll 1
        for (int h = 0; h < NPASS; ++h) {
            const int wlo = h * TPP * per, whi = min(wlo + TPP * per, n_words);
#pragma unroll
            for (int j = 0; j < WPT; ++j) {
                const int w = j * NT + tid;
                if (w >= wlo && w < whi) lds_words[w - wlo] = wreg[j];
            }
            __syncthreads();
            if (fail) {
#pragma unroll
                for (int r = 0; r < RPT; ++r) {
                    const int w = (int)((ridx[r] & ~CAND_FLAG) >> 6);
                    if (((fail >> r) & 1u) && w >= wlo && w < whi) atomicAnd(&lds_words[w - wlo], ~(1ull << (ridx[r] & 63)));
                }
            }
            __syncthreads();
            if (tid / TPP == h) {
#pragma unroll
                for (int i = 0; i < WPT; ++i) {
                    const int w = tid * per + i;
                    if (i < per && w < whi) cw[i] = lds_words[w - wlo];
                }
            }
            __syncthreads();
        }
    } else if (!mapping) {
        for (int i = tid; i < n_rec; i += NT) {
            const unsigned ri = ws.rec_idx[i];
            if (ri & CAND_FLAG) {
                bool ok = true;
                if (use_f) ok = ws.rec_q[i] < thr_f;
                if (ok && use_d) ok = ws.rec_d[i] < thr_d;
                if (!ok) atomicAnd(&ws.cand_bits[(ri & ~CAND_FLAG) >> 6], ~(1ull << (ri & 63)));
            }
        }
        __threadfence();
        __syncthreads();
    }

    KP_STAMP(3);
    // ---- ordered compaction of the candidate words
    const int w_begin = min(tid * per, n_words), w_end = min(w_begin + per, n_words);
    int cnt = 0;
    if (FAST) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) cnt += __popcll(cw[i]);
    } else {
        for (int w = w_begin; w < w_end; ++w)   // atomic loads: served by L2, where the atomics landed
            cnt += __popcll(__hip_atomic_load(&ws.cand_bits[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    int total;
    int pos = block_exclusive_scan<NT>(cnt, L.wave_tot, total);
    KP_STAMP(4);
    auto emit = [&](unsigned long long bits, int w) {
        const int row = w / words_per_row, col0 = (w - row * words_per_row) * 64;
        while (bits) {
            const int bpos = __ffsll((long long)bits) - 1;
            bits &= bits - 1;
            out_cand[pos++] = row * W + col0 + bpos;
        }
    };
    if (FAST) {
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            if (cw[i]) emit(cw[i], w_begin + i);
    } else {
        for (int w = w_begin; w < w_end; ++w)
            emit(__hip_atomic_load(&ws.cand_bits[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), w);
    }
    KP_STAMP(5);
    if (tid == 0) {
        ws.counters[0] = 0;   // leave the record counter clean for the next call (no per-call memset launch)
        out_count[0] = total;
        out_count[1] = n_rec;
        out_count[2] = 0;
        out_count[3] = 0;
        out_stats[0] = med_f;
        out_stats[1] = thr_f;
        out_stats[2] = med_d;
        out_stats[3] = thr_d;
    }
}

// RPT register slots per thread for the records, WPT for the candidate words (which pass through dynamic LDS in chunks).
// <16, 5> covers 640x480-class images (2 chunks of 20 KB), <24, 15> up to 1280x768 (8 chunks of 15 KB; 24 slots keep
// the kernel inside 128 VGPRs without scratch).
template <int NT, int RPT, int WPT, bool USE_D>
__global__ __launch_bounds__(NT) void kp_finish_kernel(mvKpSelectParams p, KpWs ws_all, int has_flow, int words_per_row,
                                                        int32_t* __restrict__ out_cand, int32_t* __restrict__ out_count,
                                                        float* __restrict__ out_stats) {
    __shared__ MedianLds L;
    extern __shared__ unsigned long long lds_words[];
    MV_CHAIN_KERNEL_PRIO();
    // lane-batched: one finishing workgroup per lane (blockIdx.x); out_cand [lanes, H*W], out_count / out_stats [lanes, 4]
    const KpWs ws = ws_all.lane(blockIdx.x);
    out_cand += (size_t)blockIdx.x * p.H * p.W;
    out_count += 4 * blockIdx.x;
    out_stats += 4 * blockIdx.x;
    const int n_rec = p.mode == MV_KP_MAPPING ? 0 : ws.counters[0];
    if (n_rec <= RPT * NT && p.H * words_per_row <= WPT * NT)
        kp_finish_body<NT, true, RPT, WPT, USE_D>(p, ws, has_flow, words_per_row, out_cand, out_count, out_stats, n_rec, L,
                                                  lds_words);
    else
        kp_finish_body<NT, false, RPT, WPT, USE_D>(p, ws, has_flow, words_per_row, out_cand, out_count, out_stats, n_rec, L,
                                                   lds_words);
}

template <int NT, int RPT, int WPT, bool USE_D>
static int launch_finish(const mvKpSelectParams& p, const KpWs& ws, int has_flow, int wpr, int32_t* out_cand,
                         int32_t* out_count, float* out_stats, int lanes, hipStream_t s) {
    const size_t dyn = (size_t)WPT * (NT / (WPT > 5 * 1024 / NT ? 8 : 2)) * sizeof(unsigned long long);   // one chunk of words
    hipLaunchKernelGGL((kp_finish_kernel<NT, RPT, WPT, USE_D>), dim3(lanes), dim3(NT), dyn, s, p, ws, has_flow, wpr, out_cand,
                       out_count, out_stats);
    return MV_OK;
}

// ---- MappingPointSelector: no medians, but typically 10^5 candidates.  The single finishing workgroup would emit ~150
// indices per thread one after the other (40 us at 640x480, 115 us at 720p); instead: one workgroup scans the per-word
// popcounts, then one WAVE per 64-pixel word emits its set bits (rank by popcount of the lower bits).
__global__ __launch_bounds__(1024) void kp_map_scan_kernel(const unsigned long long* __restrict__ cand_bits, int n_words,
                                                            int* __restrict__ word_off, int32_t* __restrict__ out_count,
                                                            float* __restrict__ out_stats) {
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x;
    const int per = (n_words + 1023) / 1024;
    const int w_begin = min(tid * per, n_words), w_end = min(w_begin + per, n_words);
    int cnt = 0;
    for (int w = w_begin; w < w_end; ++w) cnt += __popcll(cand_bits[w]);
    int total;
    int pos = block_exclusive_scan<1024>(cnt, wave_tot, total);
    for (int w = w_begin; w < w_end; ++w) {
        word_off[w] = pos;
        pos += __popcll(cand_bits[w]);
    }
    if (tid == 0) {
        out_count[0] = total;
        out_count[1] = out_count[2] = out_count[3] = 0;
        out_stats[0] = NAN;
        out_stats[1] = INFINITY;
        out_stats[2] = NAN;
        out_stats[3] = INFINITY;
    }
}

__global__ __launch_bounds__(256) void kp_map_emit_kernel(const unsigned long long* __restrict__ cand_bits,
                                                           const int* __restrict__ word_off, int n_words, int words_per_row,
                                                           int W, int32_t* __restrict__ out_cand) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_words) return;
    const int lane = threadIdx.x & 63;
    const unsigned long long bits = cand_bits[w];
    if ((bits >> lane) & 1ull) {
        const int rank = __popcll(bits & ((1ull << lane) - 1ull));
        const int row = w / words_per_row, col0 = (w - row * words_per_row) * 64;
        out_cand[word_off[w] + rank] = row * W + col0 + lane;
    }
}

__global__ void kp_gather_kernel(const int32_t* __restrict__ cand, const int64_t* __restrict__ perm, mvLaneCounts cnt,
                                 int cap, size_t cand_lane_stride, int W, int64_t* __restrict__ out_uv) {
    // lane-batched (blockIdx.y): cand [lanes, cand_lane_stride], perm / out_uv [lanes, cap(, 2)]
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_sel = cnt.n[blockIdx.y];
    cand += (size_t)blockIdx.y * cand_lane_stride;
    perm += (size_t)blockIdx.y * cap;
    out_uv += 2 * (size_t)blockIdx.y * cap;
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
    return align_up(words * 8, 256) + 3 * align_up((size_t)H * W * 4, 256) + 256;
}

static int kp_select_impl(const float* flow_cov, const float* depth0, const float* depth0_cov, const float* depth1,
                          const float* depth1_cov, const uint8_t* mask_a, const uint8_t* mask_b, const mvKpSelectParams* params,
                          void* workspace, size_t workspace_bytes, int32_t* out_cand, int32_t* out_count, float* out_stats,
                          int lanes, mvStream_t stream, const mvEpiArgs* fuse) {
    MV_CHECK_ARG(lanes >= 1 && lanes <= MV_MAX_LANES);
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
    const size_t lane_bytes = mv_kp_select_workspace_bytes(p.H, p.W);
    if (workspace_bytes < lane_bytes * (size_t)lanes) return MV_ERR_WORKSPACE;
    if (((uintptr_t)workspace & 7) != 0) return MV_ERR_INVALID_ARG;
    if (p.mode == MV_KP_MAPPING && lanes != 1) return MV_ERR_UNSUPPORTED;

    const int wpr = mv_ceil_div(p.W, 64);
    const size_t words = (size_t)p.H * wpr;
    const size_t plane_bytes = align_up((size_t)p.H * p.W * 4, 256);
    char* base = (char*)workspace;
    KpWs ws;
    ws.cand_bits = (unsigned long long*)base;
    base += align_up(words * 8, 256);
    ws.rec_idx = (unsigned*)base;
    base += plane_bytes;
    ws.rec_q = (float*)base;
    base += plane_bytes;
    ws.rec_d = (float*)base;
    base += plane_bytes;
    ws.counters = (int*)base;
    ws.lane_bytes = lane_bytes;

    hipStream_t s = (hipStream_t)stream;
    // ws.counters must be zero on entry: the caller zero-fills the workspace once after allocating it, and every
    // call leaves it zeroed again (kp_finish_kernel) — this saves a 5-us fill launch per frame.
    dim3 grid(wpr, mv_ceil_div(p.H, TILE_H), lanes), block(64, 4);
    if (fuse)
        hipLaunchKernelGGL(kp_nms_kernel<true>, grid, block, 0, s, flow_cov, depth0, depth0_cov, depth1, depth1_cov, mask_a,
                           mask_b, p, ws, wpr, *fuse);
    else
        hipLaunchKernelGGL(kp_nms_kernel<false>, grid, block, 0, s, flow_cov, depth0, depth0_cov, depth1, depth1_cov, mask_a,
                           mask_b, p, ws, wpr, mvEpiArgs{});
    if (p.mode == MV_KP_MAPPING) {
        const int n_words = p.H * wpr;
        int* word_off = (int*)ws.rec_idx;   // record arrays are unused in this mode
        hipLaunchKernelGGL(kp_map_scan_kernel, dim3(1), dim3(1024), 0, s, ws.cand_bits, n_words, word_off, out_count, out_stats);
        hipLaunchKernelGGL(kp_map_emit_kernel, dim3(mv_ceil_div(n_words, 4)), dim3(256), 0, s, ws.cand_bits, word_off, n_words,
                           wpr, p.W, out_cand);
        return mv_launch_status();
    }
    const bool big = (size_t)p.H * wpr > 5 * (size_t)1024;
    const bool full = p.mode == MV_KP_FULL;
    int rc;
    static int small_nt = -1;   // MV_KP_FINISH_SMALL_NT=512: 8-wave finishing workgroup (fits beside a 3-per-CU GEMM)
    if (small_nt < 0) { const char* e = getenv("MV_KP_FINISH_SMALL_NT"); small_nt = (e && atoi(e) == 512) ? 512 : 1024; }
    if (big) rc = full ? launch_finish<1024, 24, 15, true>(p, ws, flow_cov ? 1 : 0, wpr, out_cand, out_count, out_stats, lanes, s)
                       : launch_finish<1024, 24, 15, false>(p, ws, flow_cov ? 1 : 0, wpr, out_cand, out_count, out_stats, lanes, s);
    else if (small_nt == 512)
        rc = full ? launch_finish<512, 16, 10, true>(p, ws, flow_cov ? 1 : 0, wpr, out_cand, out_count, out_stats, lanes, s)
                  : launch_finish<512, 16, 10, false>(p, ws, flow_cov ? 1 : 0, wpr, out_cand, out_count, out_stats, lanes, s);
    else rc = full ? launch_finish<1024, 16, 5, true>(p, ws, flow_cov ? 1 : 0, wpr, out_cand, out_count, out_stats, lanes, s)
                   : launch_finish<1024, 16, 5, false>(p, ws, flow_cov ? 1 : 0, wpr, out_cand, out_count, out_stats, lanes, s);
    if (rc != MV_OK) return rc;
    return mv_launch_status();
}

extern "C" int mv_kp_select_lanes(const float* flow_cov, const float* depth0, const float* depth0_cov,
                                  const float* depth1, const float* depth1_cov, const uint8_t* mask_a,
                                  const uint8_t* mask_b, const mvKpSelectParams* params, void* workspace,
                                  size_t workspace_bytes, int32_t* out_cand, int32_t* out_count, float* out_stats,
                                  int lanes, mvStream_t stream) {
    return kp_select_impl(flow_cov, depth0, depth0_cov, depth1, depth1_cov, mask_a, mask_b, params, workspace, workspace_bytes,
                          out_cand, out_count, out_stats, lanes, stream, nullptr);
}

extern "C" int mv_frontend_epilogue_select_lanes(const float* flow, const float* logcov, int cov_is_log, float bl_fx,
                                                 float bl_fx_sq, float* disparity, float* disparity_cov, float* depth,
                                                 float* depth_cov, uint8_t* bad_mask, float* match_flow, float* match_cov,
                                                 const uint8_t* mask_a, const uint8_t* mask_b, const mvKpSelectParams* params,
                                                 void* workspace, size_t workspace_bytes, int32_t* out_cand,
                                                 int32_t* out_count, float* out_stats, int lanes, mvStream_t stream) {
    MV_CHECK_ARG(flow && logcov && params && match_cov);
    if (params->mode != MV_KP_NODEPTH) return MV_ERR_UNSUPPORTED;   // the other selectors read the PREVIOUS frame's maps as well
    const mvEpiArgs ef{flow, logcov, cov_is_log, bl_fx, bl_fx_sq, disparity, disparity_cov, depth, depth_cov, match_flow, match_cov,
                       bad_mask};
    // (flow_cov only has to be non-null for the argument check: the fused kernel never reads it)
    return kp_select_impl(match_cov, nullptr, nullptr, nullptr, nullptr, mask_a, mask_b, params, workspace, workspace_bytes,
                          out_cand, out_count, out_stats, lanes, stream, &ef);
}

extern "C" int mv_kp_select(const float* flow_cov, const float* depth0, const float* depth0_cov,
                            const float* depth1, const float* depth1_cov, const uint8_t* mask_a,
                            const uint8_t* mask_b, const mvKpSelectParams* params, void* workspace,
                            size_t workspace_bytes, int32_t* out_cand, int32_t* out_count, float* out_stats,
                            mvStream_t stream) {
    return mv_kp_select_lanes(flow_cov, depth0, depth0_cov, depth1, depth1_cov, mask_a, mask_b, params, workspace,
                              workspace_bytes, out_cand, out_count, out_stats, 1, stream);
}

extern "C" int mv_kp_gather_lanes(const int32_t* cand, size_t cand_lane_stride, const int64_t* perm, int lanes,
                                  const int32_t* n_live, int cap, int W, int64_t* out_uv, mvStream_t stream) {
    MV_CHECK_ARG(lanes >= 1 && lanes <= MV_MAX_LANES && n_live && cap >= 0 && W > 0);
    mvLaneCounts c{};
    int n_max = 0;
    for (int l = 0; l < lanes; ++l) {
        MV_CHECK_ARG(n_live[l] >= 0 && n_live[l] <= cap);
        c.n[l] = n_live[l];
        n_max = n_live[l] > n_max ? n_live[l] : n_max;
    }
    if (n_max == 0) return MV_OK;
    MV_CHECK_ARG(cand && perm && out_uv);
    hipLaunchKernelGGL(kp_gather_kernel, dim3(mv_ceil_div(n_max, 256), lanes), dim3(256), 0, (hipStream_t)stream, cand, perm,
                       c, cap, cand_lane_stride, W, out_uv);
    return mv_launch_status();
}

extern "C" int mv_kp_gather(const int32_t* cand, const int64_t* perm, int n_sel, int W, int64_t* out_uv,
                            mvStream_t stream) {
    MV_CHECK_ARG(n_sel >= 0);
    const int32_t n = n_sel;
    return mv_kp_gather_lanes(cand, 0, perm, 1, &n, n_sel, W, out_uv, stream);
}
