// A6 — (2r+1)^2 bilinear window lookup in each query pixel's own H2 x W2 cost slice
//
// Replaces FlowFormer MemoryDecoder.encode_flow_token(cost_maps, coords1) called at
// Module/Network/FlowFormerCov/covhead.py:92 ("MUST run in fp32", :91): RAFT window
// delta[i][j] = (dy[i], dx[j]) added to (x, y)  =>  channel k = K*i + j samples (x + i - r, y + j - r),
// bilinear_sampler = grid_sample(align_corners=True, zeros) after normalising by (W2-1), (H2-1).
//
// gfx950 design.  One frame's lookup (9600 queries) is ~4 MB of traffic: neither HBM nor the texture path is the
// limit, the VALU instruction stream of the few resident waves is (measured: launch floor 3.4 us, staging + stores
// +1.6 us, and the per-tap coordinate arithmetic of a lane-per-tap formulation another +3.5 us).  Hence:
//   * a wave owns QPW queries; a workgroup QPB of them (QPB consecutive queries = one contiguous output segment per
//     channel).
//   * staging: per query the (K+3)^2 cell block that covers every tap (block origin floor(x)-r-1: one spare cell each
//     side absorbs the fp32 normalise/un-normalise round trip) is fetched with lane = (row-in-group, column): the
//     lane's (row, column) split is computed once, the query origin is wave-uniform (readlane -> SGPR), so a load costs
//     a handful of VALU ops; all loads are issued before LDS is touched; out-of-image cells are zero (= zero padding).
//   * per-AXIS coordinate math: RAFT's normalise + ATen's un-normalise + floor/weights depend only on (query, i) for x
//     and (query, j) for y: 2K values per query instead of 2*K*K.  Lanes [0, 2K) of an "axis pass" evaluate them for one
//     query, the next 2K lanes for the next query, ... (true division, same fp32 op order as grid_sample) and write the
//     (weight, cell) entries to LDS.  Results are bit-identical to the per-tap form.
//   * taps (round 6): the wave's QPW * K*K (tap, query) PAIRS are dealt over its lanes (324 pairs = 6 rounds for four
//     queries; "lane = tap, query after query" took 8, every second one with 17 live lanes); a pair reads its two axis
//     entries (two 8-byte LDS reads; rounds 1-5: four ds_bpermute), 4 cells, and forms the ATen bilinear sum.
//   * output [B, K*K, H1, W1] is channel-major: results are transposed through LDS (aliasing the staging buffer) so
//     every channel row is written as one QPB*4-byte segment.
#include "common.h"

namespace {

// One axis entry of a query — what RAFT's normalise + ATen's un-normalise make of (coordinate + offset): the bilinear weight and the clamped
// cell offset in the staged block | (1 << 8) when the tap's 2-cell span lies inside the block.  The axis passes write the 2 K entries of
// every query of the wave into LDS; the tap phase reads them back per (tap, query) PAIR.
struct AxisEntry { float w; int c; };

// Tap phase shared by the lookup kernels (round 6): the wave's QPW * K * K (tap, query) pairs are dealt over its lanes — 324 pairs = 6 rounds
// of 64 for four queries where "lane = tap, one query after the other" takes 8 rounds of which every second one has 17 live lanes — and a
// pair fetches its two axis entries with two 8-byte LDS reads where the lane = tap form used four ds_bpermute.  The arithmetic per pair is
// the ATen bilinear sum of before, term for term.
template <int K, int QPW, int STRIDE, int CELLS>
__device__ __forceinline__ void tap_pairs(const float* blk, const AxisEntry (*ax)[2 * K], int lane, float (&res)[(QPW * K * K + 63) / 64]) {
    constexpr int KK = K * K, NPAIR = QPW * KK, NR = (NPAIR + 63) / 64;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int pair = min(r * 64 + lane, NPAIR - 1);
        const int s = pair / KK, tap = pair - s * KK;
        const int ti = tap / K, tj = tap - ti * K;
        const AxisEntry X = ax[s][ti], Y = ax[s][K + tj];
        const float w = X.w, n = Y.w;
        const bool inblk = ((X.c & Y.c) & 256) != 0;
        const float* p = &blk[s * CELLS + (Y.c & 255) * STRIDE + (X.c & 255)];
        const float e = 1.f - w, so = 1.f - n;
        const float vnw = inblk ? p[0] : 0.f, vne = inblk ? p[1] : 0.f;
        const float vsw = inblk ? p[STRIDE] : 0.f, vse = inblk ? p[STRIDE + 1] : 0.f;
        // ATen: (nw_val*nw + ne_val*ne) + sw_val*sw + se_val*se with nw = s*e, ne = s*w, sw = n*e, se = n*w
        float r0 = vnw * (so * e);
        r0 = r0 + vne * (so * w);
        r0 = r0 + vsw * (n * e);
        r0 = r0 + vse * (n * w);
        res[r] = r0;
    }
}

// ... and its results into the transpose buffer: outs[tap][query of the workgroup]
template <int K, int QPW, int QPB>
__device__ __forceinline__ void tap_pairs_out(float (*outs)[QPB + 1], int wave, int lane, const float (&res)[(QPW * K * K + 63) / 64]) {
    constexpr int KK = K * K, NPAIR = QPW * KK, NR = (NPAIR + 63) / 64;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int pair = r * 64 + lane;
        if (pair < NPAIR) {
            const int s = pair / KK, tap = pair - s * KK;
            outs[tap][wave * QPW + s] = res[r];
        }
    }
}

// VT = float, or _Float16 for the Fast-mode volume that is STORED in the encoder's 16-bit type (mv_corr_volume_out16: what
// `einsum` returns when the encoder runs in fp16, Config/Experiment/MACVO/MACVO_Fast.yaml:73-74; flownet.py:27 only widens it): the cells
// are widened on load — exactly `cost_maps.float()` — and everything behind the load is the fp32 arithmetic of the fp32 form.
template <int R, int QPW, int QPB, typename VT = float>
__global__ __launch_bounds__(64 * (QPB / QPW)) void corr_lookup_kernel(const VT* __restrict__ vol,
                                                                        const float* __restrict__ coords,
                                                                        float* __restrict__ out, int N1, int H2, int W2) {
    constexpr int K = 2 * R + 1;
    constexpr int KK = K * K;
    constexpr int BS = K + 3;            // staged block edge (12 for r = 4)
    constexpr int CELLS = BS * BS;       // 144
    constexpr int NWAVE = QPB / QPW;     // waves per workgroup, QPW queries each
    constexpr int NTHR = 64 * NWAVE;
    constexpr int RPR = 64 / BS;         // block rows one wave-wide load covers (5 for r = 4)
    constexpr int LPR = RPR * BS;        // active lanes of such a load (60)
    constexpr int NROUND = (BS + RPR - 1) / RPR;    // loads per query (3)
    // Round 3: the two MARGIN rows / columns of the block (index 0 and BS - 1) are read by a tap only when the fp32 normalise /
    // un-normalise round trip moves a sample across an integer, i.e. when the coordinate's fraction is within ~1e-4 of 0 (first
    // row / column) or of 1 (last).  They are fetched only then (threshold 1e-2, two orders of magnitude of slack); otherwise
    // the block's inner (BS - 2)^2 cells are everything a tap touches.  For r = 4 that is 10 rows = exactly two wave-wide loads
    // instead of three and 10-float row segments instead of 12: ~25 % fewer 32-B sectors per query for the HBM-bound batched
    // lookup (configs[4]: 882 -> ~680 B per query), one load round less for the latency-bound one.  Results unchanged bit for bit.
    constexpr bool TRIM = (BS - 2) % RPR == 0;      // inner rows are a whole number of wave-wide loads (r = 4: 10 = 2 x 5)
    constexpr int NINNER = (BS - 2) / RPR;          // ... this many
    constexpr int QPA = 64 / (2 * K);               // queries one axis pass covers (3 for r = 4)
    constexpr int APASS = (QPW + QPA - 1) / QPA;
    static_assert(QPW <= 32 && (QPW & (QPW - 1)) == 0 && QPB % QPW == 0, "bad lookup tiling");

    // one LDS region, two lives: staged cell blocks (read by the tap phase), then the transposed outputs
    constexpr int BLK_STRIDE = QPW * CELLS;
    constexpr int BLK_FLOATS = NWAVE * BLK_STRIDE;
    constexpr int OUT_FLOATS = KK * (QPB + 1);
    __shared__ float smem[BLK_FLOATS > OUT_FLOATS ? BLK_FLOATS : OUT_FLOATS];
    float* blk = smem + (threadIdx.x >> 6) * BLK_STRIDE;
    float (*outs)[QPB + 1] = reinterpret_cast<float (*)[QPB + 1]>(smem);

    MV_SMALL_KERNEL_PRIO();
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * QPB;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int slice = H2 * W2;

    // lane s < QPW owns query s of this wave: load its coords, derive the block origin
    const int qmine = q0 + wave * QPW + (lane & (QPW - 1));
    float x = 0.f, y = 0.f;
    if (qmine < N1) {
        x = coords[((size_t)b * 2 + 0) * N1 + qmine];
        y = coords[((size_t)b * 2 + 1) * N1 + qmine];
    }
    // clamp only the integer origin (NaN / huge coords must not produce wild addresses)
    const float xc = fminf(fmaxf(x, -1.0e6f), 1.0e6f), yc = fminf(fmaxf(y, -1.0e6f), 1.0e6f);
    const int bx = ((xc == xc) ? (int)floorf(xc) : 0) - R - 1;
    const int by = ((yc == yc) ? (int)floorf(yc) : 0) - R - 1;
    // margin flags of this lane's query: bit 0 first column, 1 last column, 2 first row, 3 last row (NaN: none — all taps are zero)
    const float frx = xc - floorf(xc), fry = yc - floorf(yc);
    const int margin = ((frx < 0.01f) ? 1 : 0) | ((frx > 0.99f) ? 2 : 0) | ((fry < 0.01f) ? 4 : 0) | ((fry > 0.99f) ? 8 : 0);

    // ---- stage QPW x CELLS cells: issue every load before touching LDS
    const int cyl = lane / BS, cxl = lane - cyl * BS;
    float v[QPW][NROUND];
#pragma unroll
    for (int s = 0; s < QPW; ++s) {
        const int sbx = __builtin_amdgcn_readlane(bx, s), sby = __builtin_amdgcn_readlane(by, s);
        const int q = q0 + wave * QPW + s;
        const VT* __restrict__ base = vol + ((size_t)b * N1 + q) * slice;
        const int gx = sbx + cxl;
        if (TRIM) {
            const int mg = __builtin_amdgcn_readlane(margin, s);                 // wave-uniform
            const bool okx = lane < LPR && q < N1 && gx >= 0 && gx < W2 && (cxl != 0 || (mg & 1)) && (cxl != BS - 1 || (mg & 2));
#pragma unroll
            for (int k = 0; k < NINNER; ++k) {                                   // inner rows 1 .. BS - 2
                const int row = 1 + k * RPR + cyl, gy = sby + row;
                const bool ok = okx && gy >= 0 && gy < H2;
                v[s][k] = ok ? (float)base[gy * W2 + gx] : 0.f;
            }
            v[s][NROUND - 1] = 0.f;
            if (mg & 12) {                                                       // rare: the margin rows, lanes [0, BS) row 0, [BS, 2 BS) row BS - 1
                const int row = cyl == 0 ? 0 : BS - 1, gy = sby + row;
                const bool ok = okx && cyl < 2 && ((mg >> (cyl == 0 ? 2 : 3)) & 1) && gy >= 0 && gy < H2;
                v[s][NROUND - 1] = ok ? (float)base[gy * W2 + gx] : 0.f;
            }
        } else {
            const bool okx = lane < LPR && q < N1 && gx >= 0 && gx < W2;
#pragma unroll
            for (int k = 0; k < NROUND; ++k) {
                const int row = k * RPR + cyl, gy = sby + row;
                const bool ok = okx && row < BS && gy >= 0 && gy < H2;
                v[s][k] = ok ? (float)base[gy * W2 + gx] : 0.f;
            }
        }
    }

    // ---- per-axis coordinate math (overlaps the loads in flight): lane -> (query sq of the pass, axis, offset)
    const float wm1 = (float)(W2 - 1), hm1 = (float)(H2 - 1);
    const int asq = lane / (2 * K), aa = lane - asq * (2 * K);
    const bool a_is_y = aa >= K;
    const int aoff = (a_is_y ? aa - K : aa) - R;
    const float adim = a_is_y ? hm1 : wm1;
    __shared__ AxisEntry ax_all[NWAVE][QPW][2 * K];
    AxisEntry (*ax)[2 * K] = ax_all[wave];
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps) {
        const int s = ps * QPA + asq;   // lanes with s >= QPW compute garbage nobody keeps
        const float qx = __shfl(x, s, 64), qy = __shfl(y, s, 64);
        const int obx = __shfl(bx, s, 64), oby = __shfl(by, s, 64);
        const float cs = (a_is_y ? qy : qx) + (float)aoff;
        const float g = (2.f * cs) / adim - 1.f;          // RAFT bilinear_sampler normalisation
        const float ic = (g + 1.f) * (adim / 2.f);        // ATen grid_sampler_unnormalize, align_corners=True
        const float f0 = floorf(ic);
        const int c = (int)fminf(fmaxf(f0, -2.0e6f), 2.0e6f) - (a_is_y ? oby : obx);
        const bool inb = c >= 0 && c <= BS - 2;
        if (asq < QPA && s < QPW) ax[s][aa] = AxisEntry{ic - f0, min(max(c, 0), BS - 2) | (inb ? 256 : 0)};
    }

#pragma unroll
    for (int s = 0; s < QPW; ++s) {
        if (TRIM) {
#pragma unroll
            for (int k = 0; k < NINNER; ++k)
                if (lane < LPR) blk[s * CELLS + BS + k * LPR + lane] = v[s][k];                        // rows 1 + 5 k ..
            if (lane < 2 * BS) blk[s * CELLS + (lane < BS ? lane : (BS - 2) * BS + lane)] = v[s][NROUND - 1];   // rows 0 and BS - 1 (zeros unless fetched)
        } else {
#pragma unroll
            for (int k = 0; k < NROUND; ++k)
                if (lane < LPR && k * LPR + lane < CELLS) blk[s * CELLS + k * LPR + lane] = v[s][k];
        }
    }
    __syncthreads();

    // ---- taps: (tap, query) pairs over the lanes
    float res[(QPW * KK + 63) / 64];
    tap_pairs<K, QPW, BS, CELLS>(blk, ax, lane, res);
    __syncthreads();   // every wave is done reading the staged blocks: the region becomes the output transpose buffer
    tap_pairs_out<K, QPW, QPB>(outs, wave, lane, res);
    __syncthreads();

    // ---- transposed store: each channel row = QPB consecutive queries
    for (int idx = t; idx < KK * QPB; idx += NTHR) {
        const int k = idx / QPB, c = idx - k * QPB;
        if (q0 + c < N1) out[((size_t)b * KK + k) * N1 + q0 + c] = outs[k][c];
    }
}


// ---- the same lookup in a TILED volume (round 3): query q's slice is stored as 4 x 4-cell tiles,
//      vol[(b N1 + q) H2 W2 + (ty * W2/4 + tx) * 16 + (y % 4) * 4 + x % 4]   — what mv_corr_volume_packed writes when operand 2 was packed by
// mv_volume_pack_tiled (the GEMM itself is unchanged: only the order of its columns is).  A tile is one aligned 64-byte line, so the
// 10 x 10 (12 x 12 with margins) cell block of a query costs (10 + 3) / 4 = 3.25 tiles per axis = ~10.6 lines = 676 B instead of
// 10 row segments x 2.1 sectors of 32 B = 680 B in 21 separately addressed pieces: the same bytes in half as many, aligned, fully
// used DRAM bursts.  Staging: a wave-wide load = one tile row of the 4 x 4 tile grid that covers the block (lane = tile column x 16
// cells); tiles outside the needed cell range or the image are skipped / zero.  The staged block is 16 x 16 cells anchored at the
// first tile; axis math, tap phase and the transposed store are those of corr_lookup_kernel (r = 4 only), results bit-identical.
//
// VT = _Float16 (round 6): the Fast-mode volume's 2-byte cells in the same tile order (mv_fmap_tile_rows16 permutes operand 2's
// pixel rows in front of mv_corr_volume_out16).  A tile is then ONE aligned 32-byte sector: a query's block costs ~10.6 sectors = 338 B where
// the row-major fp16 volume costs 10 row segments of 20 B that straddle sectors (1.6 each: ~512 B).  Staging: a lane fetches two
// neighbouring cells as one dword, so a wave-wide load covers two tile rows and a query needs two loads; the cells are widened on load
// (`cost_maps.float()`, exact) and everything behind the load is the fp32 arithmetic above — tokens bit-identical to
// mv_corr_lookup_vol16 on the row-major volume.
template <int QPW, int QPB, typename VT = float>
__global__ __launch_bounds__(64 * (QPB / QPW)) void corr_lookup_tiled_kernel(const VT* __restrict__ vol,
                                                                              const float* __restrict__ coords,
                                                                              float* __restrict__ out, int N1, int H2, int W2) {
    constexpr int R = 4, K = 9, KK = 81;
    constexpr int BS = 16;                     // staged block: 4 x 4 tiles of 4 x 4 cells ...
    constexpr int BSP = 18, CELLS = BS * BSP;  // ... at a row stride of 18 floats: the 9 rows a wave's taps read sit in 9 different banks (16: rows y, y + 2 collide)
    constexpr int NWAVE = QPB / QPW, NTHR = 64 * NWAVE;
    constexpr int QPA = 64 / (2 * K);
    constexpr int APASS = (QPW + QPA - 1) / QPA;
    constexpr int BLK_STRIDE = QPW * CELLS;
    constexpr int BLK_FLOATS = NWAVE * BLK_STRIDE;
    constexpr int OUT_FLOATS = KK * (QPB + 1);
    __shared__ float smem[BLK_FLOATS > OUT_FLOATS ? BLK_FLOATS : OUT_FLOATS];
    float* blk = smem + (threadIdx.x >> 6) * BLK_STRIDE;
    float (*outs)[QPB + 1] = reinterpret_cast<float (*)[QPB + 1]>(smem);

    MV_SMALL_KERNEL_PRIO();
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * QPB;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tpr = W2 >> 2, tpc = (H2 + 3) >> 2, slice = tpc * tpr * 16;   // (fp16 cells: H2 need not be a multiple of 4 — the last tile row is padded with zero cells)

    const int qmine = q0 + wave * QPW + (lane & (QPW - 1));
    float x = 0.f, y = 0.f;
    if (qmine < N1) {
        x = coords[((size_t)b * 2 + 0) * N1 + qmine];
        y = coords[((size_t)b * 2 + 1) * N1 + qmine];
    }
    const float xc = fminf(fmaxf(x, -1.0e6f), 1.0e6f), yc = fminf(fmaxf(y, -1.0e6f), 1.0e6f);
    const int bx = ((xc == xc) ? (int)floorf(xc) : 0) - R - 1;      // origin of the 12 x 12 cell block of corr_lookup_kernel
    const int by = ((yc == yc) ? (int)floorf(yc) : 0) - R - 1;
    const float frx = xc - floorf(xc), fry = yc - floorf(yc);
    const int x_lo = bx + ((frx < 0.01f) ? 0 : 1), x_hi = bx + ((frx > 0.99f) ? K + 2 : K + 1);   // cells a tap can touch (see TRIM above)
    const int y_lo = by + ((fry < 0.01f) ? 0 : 1), y_hi = by + ((fry > 0.99f) ? K + 2 : K + 1);
    const int tox = bx >> 2, toy = by >> 2;                          // first tile (floor division: arithmetic shift)

    // ---- stage.  fp32 cells: per query four wave-wide loads (tile rows), lane = (tile column, cell).  fp16 cells: two loads of two tile
    //      rows each, lane = (tile row of the pair, tile column, cell pair)
    constexpr bool H16 = sizeof(VT) == 2;
    constexpr int NLD = H16 ? 2 : 4;
    const int ti = H16 ? (lane >> 3) & 3 : lane >> 4, cell = H16 ? (lane & 7) * 2 : lane & 15;
    const int half_row = H16 ? lane >> 5 : 0;
    float v[QPW][4];
#pragma unroll
    for (int s = 0; s < QPW; ++s) {
        const int stx = __builtin_amdgcn_readlane(tox, s), sty = __builtin_amdgcn_readlane(toy, s);
        const int sxl = __builtin_amdgcn_readlane(x_lo, s), sxh = __builtin_amdgcn_readlane(x_hi, s);
        const int syl = __builtin_amdgcn_readlane(y_lo, s), syh = __builtin_amdgcn_readlane(y_hi, s);
        const int q = q0 + wave * QPW + s;
        const VT* __restrict__ base = vol + ((size_t)b * N1 + q) * slice;
        const int tx = stx + ti;
        const bool okx = q < N1 && tx >= 0 && tx < tpr && 4 * tx + 3 >= sxl && 4 * tx <= sxh;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int ty = sty + (H16 ? 2 * k + half_row : k);
            const bool ok = okx && ty >= 0 && ty < tpc && 4 * ty + 3 >= syl && 4 * ty <= syh;
            if constexpr (H16) {
                union { uint32_t u; _Float16 h[2]; } two;
                two.u = ok ? *reinterpret_cast<const uint32_t*>(base + ((ty * tpr + tx) << 4) + cell) : 0u;
                v[s][2 * k] = (float)two.h[0];
                v[s][2 * k + 1] = (float)two.h[1];
            } else {
                v[s][k] = ok ? (float)base[((ty * tpr + tx) << 4) + cell] : 0.f;
            }
        }
    }

    // ---- per-axis coordinate math (as corr_lookup_kernel; cell offsets relative to the first tile)
    const float wm1 = (float)(W2 - 1), hm1 = (float)(H2 - 1);
    const int asq = lane / (2 * K), aa = lane - asq * (2 * K);
    const bool a_is_y = aa >= K;
    const int aoff = (a_is_y ? aa - K : aa) - R;
    const float adim = a_is_y ? hm1 : wm1;
    __shared__ AxisEntry ax_all[NWAVE][QPW][2 * K];
    AxisEntry (*ax)[2 * K] = ax_all[wave];
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps) {
        const int s = ps * QPA + asq;
        const float qx = __shfl(x, s, 64), qy = __shfl(y, s, 64);
        const int obx = __shfl(tox, s, 64) * 4, oby = __shfl(toy, s, 64) * 4;
        const float cs = (a_is_y ? qy : qx) + (float)aoff;
        const float g = (2.f * cs) / adim - 1.f;
        const float ic = (g + 1.f) * (adim / 2.f);
        const float f0 = floorf(ic);
        const int c = (int)fminf(fmaxf(f0, -2.0e6f), 2.0e6f) - (a_is_y ? oby : obx);
        const bool inb = c >= 0 && c <= BS - 2;
        if (asq < QPA && s < QPW) ax[s][aa] = AxisEntry{ic - f0, min(max(c, 0), BS - 2) | (inb ? 256 : 0)};
    }

#pragma unroll
    for (int s = 0; s < QPW; ++s)
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            if constexpr (H16)      // the pair sits in one cell row of the tile, at an even column: one 8-byte LDS write
                *reinterpret_cast<float2*>(&blk[s * CELLS + (4 * (2 * k + half_row) + (cell >> 2)) * BSP + 4 * ti + (cell & 3)]) =
                    make_float2(v[s][2 * k], v[s][2 * k + 1]);
            else
                blk[s * CELLS + (4 * k + (cell >> 2)) * BSP + 4 * ti + (cell & 3)] = v[s][k];
        }
    __syncthreads();

    float res[(QPW * KK + 63) / 64];
    tap_pairs<K, QPW, BSP, CELLS>(blk, ax, lane, res);
    __syncthreads();
    tap_pairs_out<K, QPW, QPB>(outs, wave, lane, res);
    __syncthreads();
    for (int idx = t; idx < KK * QPB; idx += NTHR) {
        const int k = idx / QPB, c = idx - k * QPB;
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
    hipStream_t s = (hipStream_t)stream;
    // small launches (one frame: 9600 queries for 256 CUs) are latency-bound -> 16-query workgroups of 8 waves (600
    // workgroups, 5.96 us vs 7.15 us for 32-query ones); large batches (B = 32: 58.5 us, ~2.4 TB/s) are throughput-bound ->
    // 4 queries per wave amortise the per-wave setup, 32-query (128-B) output segments
    const bool small = (size_t)B * N1 <= (size_t)lookup_small_threshold();
    static int small_qpb = -1;   // MV_LOOKUP_QPB=8: 256-thread workgroups of 8 queries (A/B knob for the co-running case)
    if (small_qpb < 0) { const char* e = getenv("MV_LOOKUP_QPB"); small_qpb = (e && atoi(e) == 8) ? 8 : 16; }
#define MV_LOOKUP(R)                                                                                                  \
    if (small && small_qpb == 8)                                                                                      \
        hipLaunchKernelGGL((corr_lookup_kernel<R, 2, 8>), dim3(mv_ceil_div(N1, 8), B), dim3(256), 0, s, vol, coords,  \
                           out, N1, H2, W2);                                                                          \
    else if (small)                                                                                                   \
        hipLaunchKernelGGL((corr_lookup_kernel<R, 2, 16>), dim3(mv_ceil_div(N1, 16), B), dim3(512), 0, s, vol, coords, \
                           out, N1, H2, W2);                                                                          \
    else                                                                                                              \
        hipLaunchKernelGGL((corr_lookup_kernel<R, 4, 32>), dim3(mv_ceil_div(N1, 32), B), dim3(512), 0, s, vol, coords, \
                           out, N1, H2, W2)
    switch (radius) {
        case 1: MV_LOOKUP(1); break;
        case 2: MV_LOOKUP(2); break;
        case 3: MV_LOOKUP(3); break;
        default: MV_LOOKUP(4); break;
    }
#undef MV_LOOKUP
    return mv_launch_status();
}

// the same lookup on a volume stored as fp16 (mv_corr_volume_out16): half the bytes per cell; tokens are fp32
extern "C" int mv_corr_lookup_vol16(const void* vol, const float* coords, float* out, int B, int H1, int W1, int H2, int W2, int radius,
                                    mvStream_t stream) {
    MV_CHECK_ARG(vol && coords && out);
    MV_CHECK_ARG(B > 0 && H1 > 0 && W1 > 0 && H2 > 1 && W2 > 1);
    if (radius != 4 || B > 65535) return MV_ERR_UNSUPPORTED;
    const int N1 = H1 * W1;
    hipStream_t s = (hipStream_t)stream;
    const _Float16* v = reinterpret_cast<const _Float16*>(vol);
    if ((size_t)B * N1 <= (size_t)lookup_small_threshold())
        hipLaunchKernelGGL((corr_lookup_kernel<4, 2, 16, _Float16>), dim3(mv_ceil_div(N1, 16), B), dim3(512), 0, s, v, coords, out, N1, H2, W2);
    else
        hipLaunchKernelGGL((corr_lookup_kernel<4, 4, 32, _Float16>), dim3(mv_ceil_div(N1, 32), B), dim3(512), 0, s, v, coords, out, N1, H2, W2);
    return mv_launch_status();
}

// the same on a volume whose slices are stored in 4 x 4-cell tiles (mv_volume_pack_tiled + mv_corr_volume_packed); radius 4, H2 and W2
// multiples of 4
extern "C" int mv_corr_lookup_tiled(const float* vol, const float* coords, float* out, int B, int H1, int W1, int H2, int W2,
                                    int radius, mvStream_t stream) {
    MV_CHECK_ARG(vol && coords && out);
    MV_CHECK_ARG(B > 0 && H1 > 0 && W1 > 0 && H2 > 1 && W2 > 1);
    if (radius != 4 || (H2 % 4) || (W2 % 4) || B > 65535) return MV_ERR_UNSUPPORTED;
    const int N1 = H1 * W1;
    hipStream_t s = (hipStream_t)stream;
    if ((size_t)B * N1 <= (size_t)lookup_small_threshold())
        hipLaunchKernelGGL((corr_lookup_tiled_kernel<2, 16>), dim3(mv_ceil_div(N1, 16), B), dim3(512), 0, s, vol, coords, out, N1, H2, W2);
    else
        hipLaunchKernelGGL((corr_lookup_tiled_kernel<4, 32>), dim3(mv_ceil_div(N1, 32), B), dim3(512), 0, s, vol, coords, out, N1, H2, W2);
    return mv_launch_status();
}


// ... and on a TILED volume of fp16 cells: mv_fmap_tile_rows16 on operand 2 + mv_corr_volume_out16 (radius 4, H2 and W2 multiples of 4)
extern "C" int mv_corr_lookup_tiled_vol16(const void* vol, const float* coords, float* out, int B, int H1, int W1, int H2, int W2, int radius,
                                          mvStream_t stream) {
    MV_CHECK_ARG(vol && coords && out);
    MV_CHECK_ARG(B > 0 && H1 > 0 && W1 > 0 && H2 > 1 && W2 > 1);
    MV_CHECK_ARG(((uintptr_t)vol & 31) == 0);                            // a tile = one 32-byte sector
    if (radius != 4 || (W2 % 4) || B > 65535) return MV_ERR_UNSUPPORTED;
    const int N1 = H1 * W1;
    hipStream_t s = (hipStream_t)stream;
    const _Float16* v = reinterpret_cast<const _Float16*>(vol);
    if ((size_t)B * N1 <= (size_t)lookup_small_threshold())
        hipLaunchKernelGGL((corr_lookup_tiled_kernel<2, 16, _Float16>), dim3(mv_ceil_div(N1, 16), B), dim3(512), 0, s, v, coords, out, N1, H2, W2);
    else
        hipLaunchKernelGGL((corr_lookup_tiled_kernel<4, 32, _Float16>), dim3(mv_ceil_div(N1, 32), B), dim3(512), 0, s, v, coords, out, N1, H2, W2);
    return mv_launch_status();
}

// The pixel rows of a 16-bit HWC feature map [B, H, W, C] in 4 x 4-tile order: out[b][(ty * W/4 + tx) * 16 + (y % 4) * 4 + x % 4][:] =
// f[b][y * W + x][:], ceil(H / 4) tile rows — rows y >= H of the last one are ZERO pixels, so out is [B, mv_tiled_slice_cells(H, W), C].  As
// operand 2 of mv_corr_volume_out16 (N2 = that many) it makes the unchanged GEMM write every query's slice tiled (each output element is the
// same k-ordered sum wherever its column sits; the padding cells come out as exact zeros = the lookup's zero padding) — the 16-bit twin of
// mv_volume_pack_tiled's permutation.  One 16-byte chunk per thread.
namespace {
__global__ __launch_bounds__(256) void fmap_tile_rows16_kernel(const uint4* __restrict__ f, uint4* __restrict__ out, int H, int W, int n_out, int cpr, size_t total) {
    const int tpr = W >> 2;
    const size_t n_in = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t rowi = i / cpr;
        const int ch = (int)(i - rowi * cpr);
        const size_t b = rowi / n_out;
        const int j = (int)(rowi - b * n_out);
        const int tile = j >> 4, cell = j & 15;
        const int ty = tile / tpr, tx = tile - ty * tpr;
        const int y = 4 * ty + (cell >> 2);
        out[i] = y < H ? f[(b * n_in + (size_t)y * W + 4 * tx + (cell & 3)) * cpr + ch] : make_uint4(0u, 0u, 0u, 0u);
    }
}
}  // namespace

extern "C" int mv_tiled_slice_cells(int H, int W) { return (H > 0 && W > 0 && (W % 4) == 0) ? ((H + 3) / 4) * 4 * W : 0; }

extern "C" int mv_fmap_tile_rows16(const void* f, void* out, int B, int C, int H, int W, mvStream_t stream) {
    MV_CHECK_ARG(f && out && f != out && B > 0 && C > 0 && H > 0 && W > 0);
    MV_CHECK_ARG(((uintptr_t)f & 15) == 0 && ((uintptr_t)out & 15) == 0);
    if ((C % 8) || (W % 4)) return MV_ERR_UNSUPPORTED;
    const int cpr = C / 8, n_out = mv_tiled_slice_cells(H, W);
    const size_t total = (size_t)B * n_out * cpr;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(fmap_tile_rows16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)f, (uint4*)out, H, W, n_out, cpr, total);
    return mv_launch_status();
}
