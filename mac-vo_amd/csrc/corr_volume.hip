// A5 — all-pairs cost volume  out[b,i,j] = sum_c f1[b,c,i] * f2[b,c,j]   
//
// Replaces FlowFormer MemoryEncoder.corr (einsum 'bhid,bhjd->bhij', heads = 1, no 1/sqrt(d)) called at
// Module/Network/FlowFormerCov/flownet.py:26 and the `.float()` at :27.
//
// gfx950 design
//   * fp32 inputs: exact-fp32 MFMA v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain, k ascending).  The NCHW
//     feature map is K-major for BOTH operands ([C][N]), which is exactly the 32x32x2 A/B fragment order
//     (lane l holds element [k = l>>5][i = l&31]) — no transposition anywhere: global float4 rows ->
//     LDS [BK][128] -> conflict-free ds_read_b32 -> MFMA.  MFMA-bound (AI ~116 FLOP/B).
//   * 16-bit inputs (Fast mode): v_mfma_f32_32x32x16_{f16,bf16}, fp32 accumulate, fp32 out — HBM-write bound.
//   * 128x128 output tile per 256-thread workgroup (4 waves, 64x64 each = 2x2 MFMA blocks, 64 acc VGPRs),
//     double-buffered LDS, register-staged global prefetch.
//   * stores: each v_mfma C register row is 32 consecutive floats = one full 128-B line per half-wave and instruction.
//     (Measured alternative: swapping the MFMA operands gives each lane 4 consecutive columns -> 16 dwordx4 stores
//     instead of 64 dword stores per wave, but every instruction then writes 32 B into 32 different lines: the
//     16-bit kernel went 59 -> 112 us, the fp32 kernel 250 -> 279 us.  Full-line stores win.  An LDS-staged epilogue
//     with dwordx4 stores of whole 256-B row segments measured the same as the direct one (59 us), so the 16-bit
//     kernel's distance to the 27-30 us of a pure 184-MB fill is not store-instruction count; BK 32 vs 64 and nt vs
//     plain stores are also within noise.  An LDS-free fp32 variant (fragments straight from global memory with an
//     8- or 16-deep register ring, no barriers) measured 395 / 628 us vs 250 us: the LDS tile is worth keeping.)
#include "common.h"
#include "vol_asm.h"
#include <atomic>
#include <type_traits>
#include <stdlib.h>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BM = 128;
constexpr int BN = 128;
#ifdef MV_VOL_RG_RUNTIME
__constant__ int g_rg = 5;   // experiment knob: tile rows per super-row (scratch builds only)
#endif

// XCD-aware tile order: consecutive workgroup ids land on different XCDs (id % 8); give each XCD a
// contiguous band of tile rows so the f1 row-band and the streamed f2 tiles stay in that XCD's L2.
__device__ __forceinline__ void tile_coords(int tiles_m, int tiles_n, int& tm, int& tn, bool remap = true,
                                            int vid = -1) {
    const int nwg = tiles_m * tiles_n;
    const int id = vid >= 0 ? vid : (int)blockIdx.x;
    if (!remap) {
        tm = id / tiles_n;
        tn = id - tm * tiles_n;
        return;
    }
    const int xcd = id & 7;
    const int q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int lin = base + (id >> 3);
    // inside an XCD's contiguous range, walk "super-rows" of RG tile rows column by column: consecutive workgroups share
    // one f2 column tile (128 KB) and cycle through RG f1 row tiles (RG x 128 KB = 1.3 MB, resident in the 4 MB L2).
    // With RG = 10 a super-row holds 380 tiles = two XCDs' worth (361 per pair / 8 ... 180 each), i.e. the XCDs tile the
    // output 4 x 2: each reads 1/4 of f1 and 1/2 of f2 instead of 1/8 and all of it (HBM/fabric fetches 100 -> ~60 MB;
    // the run time does not depend on RG — the GEMM is not read-bound)
#ifdef MV_VOL_RG_RUNTIME
    const int RG = g_rg;
#else
    constexpr int RG = 10;
#endif
    const int per_sr = RG * tiles_n;
    const int sr = lin / per_sr;
    const int within = lin - sr * per_sr;
    const int rows = min(RG, tiles_m - sr * RG);
    tn = within / rows;
    tm = sr * RG + (within - tn * rows);
}


// ---- shared pieces of the fp32 kernels -----------------------------------------------------------------
// One K-tile (BK k values) of the 64x64 per-wave product from K-major LDS tiles sA/sB ([BK][LD] floats).
// Fragments of k-pair kk+2 are fetched before the MFMAs of k-pair kk are issued, so the LDS latency sits
// under 4 x 64 cycles of matrix work instead of in front of it.
template <int BK, int LD>
__device__ __forceinline__ void mfma_tile_f32(const float (*__restrict__ sA)[LD], const float (*__restrict__ sB)[LD],
                                              int rowA, int rowB, int kh, f32x16 (&acc)[2][2]) {
    float a[2][2], b[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a[0][i] = sA[kh][rowA + i * 32];
        b[0][i] = sB[kh][rowB + i * 32];
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
        const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
        if (kk + 2 < BK) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[nxt][i] = sA[kk + 2 + kh][rowA + i * 32];
                b[nxt][i] = sB[kk + 2 + kh][rowB + i * 32];
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMAs (hipcc otherwise sinks it back)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Interior tiles take the
// unguarded path (one branch per workgroup instead of one per element).
template <bool NT = true>
__device__ __forceinline__ void store_tile(float* __restrict__ O, const f32x16 (&acc)[2][2], int row0, int col0,
                                           int kh, int li, int N1, int N2, bool interior) {
    if (interior) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* p = O + (size_t)(row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * N2 + col0 + li;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (NT) MV_VOL_STORE(acc[i][j][r], p + j * 32);
                    else p[j * 32] = acc[i][j][r];
                }
            }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < N1) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int col = col0 + j * 32 + li;
                        if (col < N2) MV_VOL_STORE(acc[i][j][r], &O[(size_t)row * N2 + col]);
                    }
                }
            }
    }
}


// ------------------------------------------------------------------------------------------------
// fp32, CHW ([C][N]) operands.  BK = 16.
//
// PERSIST form (opt-in, see persistent_grid()): the grid is k workgroups per CU (768 on MI355X, a multiple of 8 so that
// "id % 8 == XCD" keeps holding) and each workgroup strides over the B * tiles_m * tiles_n tiles.  The whole grid is
// resident from the first microsecond, so the hardware workgroup dispatcher is never parked on this kernel, and the
// small latency-bound kernels that the pipeline runs on other streams (window lookups, selector, backend, PGO of the
// neighbouring frames) are dispatched into the spare wave slots / LDS while the GEMM runs.  With the classic
// one-tile-per-workgroup grid (2888 workgroups for 1024 slots) a concurrently launched 8-us lookup kernel was measured
// to finish only when the GEMM had drained (140 us) — head-of-line blocking in the dispatcher, not lack of resources.
// ------------------------------------------------------------------------------------------------
template <bool VEC4, bool PERSIST>
__global__ __launch_bounds__(256) void corr_volume_f32_chw(const float* __restrict__ f1,
                                                            const float* __restrict__ f2,
                                                            float* __restrict__ out, int C, int N1, int N2,
                                                            int tiles_m, int tiles_n, int batches) {
    constexpr int BK = 16;
    constexpr int NP = BK / 8;
    __shared__ __attribute__((aligned(16))) float sA[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float sB[2][BK][BN];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int kh = lane >> 5;        // which k of the pair this lane feeds
    const int li = lane & 31;
    // loader mapping: BK rows x 128 cols per operand = BK*32 float4; thread t takes (row = t/32 + 8*p, col4 = (t%32)*4)
    const int lrow = t >> 5;
    const int lcol = (t & 31) * 4;
    const int nk = C / BK;
    const int per_batch = tiles_m * tiles_n;
    const int total = PERSIST ? per_batch * batches : 1;

    for (int work = PERSIST ? (int)blockIdx.x : 0; work < total; work += PERSIST ? (int)gridDim.x : 1) {
        int tm, tn, b;
        if (PERSIST) {
            // batch-major split keeps "virtual id % 8 == XCD": gridDim.x is a multiple of 8
            b = work / per_batch;
            tile_coords(tiles_m, tiles_n, tm, tn, true, work - b * per_batch);
        } else {
            b = blockIdx.z;
            tile_coords(tiles_m, tiles_n, tm, tn);
        }
        const int m0 = tm * BM, n0 = tn * BN;
        const float* A = f1 + (size_t)b * C * N1;
        const float* Bp = f2 + (size_t)b * C * N2;

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        if (VEC4 && (nk & 1) == 0) {
            // 2-deep global prefetch, register staged: the loads of tile kt+2 are issued while tile kt is multiplied, so the
            // wait in front of the LDS store of tile kt+1 is vmcnt(4) — the younger loads stay in flight across the
            // barrier — instead of the vmcnt(0) drain of the 1-deep form (measured 216.8 -> 207.7 us).  For hipcc to count
            // them the loads must be branch-free: columns past the edge are clamped to the last valid float4 (they only
            // feed output rows / columns that are never stored) and the last two issues re-read the final tile.
            const float* pa = A + min(m0 + lcol, N1 - 4);
            const float* pb = Bp + min(n0 + lcol, N2 - 4);
            const int klast = C - BK;
            f32x4 ra[2][NP], rb[2][NP];
            auto gload = [&](auto SET, int k0) {
                constexpr int S = decltype(SET)::value;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const int k = k0 + lrow + 8 * p;
                    ra[S][p] = *reinterpret_cast<const f32x4*>(pa + (size_t)k * N1);
                    rb[S][p] = *reinterpret_cast<const f32x4*>(pb + (size_t)k * N2);
                }
            };
            auto sstore = [&](auto SET, int buf) {
                constexpr int S = decltype(SET)::value;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    *reinterpret_cast<f32x4*>(&sA[buf][lrow + 8 * p][lcol]) = ra[S][p];
                    *reinterpret_cast<f32x4*>(&sB[buf][lrow + 8 * p][lcol]) = rb[S][p];
                }
            };
            using S0 = std::integral_constant<int, 0>;
            using S1 = std::integral_constant<int, 1>;
            gload(S0{}, 0);
            gload(S1{}, BK);
            if (PERSIST) __syncthreads();   // the previous tile's last K-step may still be reading buffer 0/1
            sstore(S0{}, 0);
            __syncthreads();
            for (int kt = 0; kt < nk; kt += 2) {
                // even step: tile kt in LDS buffer 0, tile kt+1 in register set 1
                gload(S0{}, min((kt + 2) * BK, klast));
                mfma_tile_f32<BK, BM>(sA[0], sB[0], wm * 64 + li, wn * 64 + li, kh, acc);
                sstore(S1{}, 1);
                __syncthreads();
                // odd step: tile kt+1 in LDS buffer 1, tile kt+2 in register set 0
                gload(S1{}, min((kt + 3) * BK, klast));
                mfma_tile_f32<BK, BM>(sA[1], sB[1], wm * 64 + li, wn * 64 + li, kh, acc);
                sstore(S0{}, 0);
                __syncthreads();
            }
        } else {
            f32x4 ra[NP], rb[NP];
            auto gload = [&](int k0) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const int k = k0 + lrow + 8 * p;
                    const float* pa = A + (size_t)k * N1 + m0 + lcol;
                    const float* pb = Bp + (size_t)k * N2 + n0 + lcol;
                    if (VEC4) {
                        ra[p] = (m0 + lcol < N1) ? *reinterpret_cast<const f32x4*>(pa) : f32x4{0, 0, 0, 0};
                        rb[p] = (n0 + lcol < N2) ? *reinterpret_cast<const f32x4*>(pb) : f32x4{0, 0, 0, 0};
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            ra[p][e] = (m0 + lcol + e < N1) ? pa[e] : 0.f;
                            rb[p][e] = (n0 + lcol + e < N2) ? pb[e] : 0.f;
                        }
                    }
                }
            };
            auto sstore = [&](int buf) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    *reinterpret_cast<f32x4*>(&sA[buf][lrow + 8 * p][lcol]) = ra[p];
                    *reinterpret_cast<f32x4*>(&sB[buf][lrow + 8 * p][lcol]) = rb[p];
                }
            };
            gload(0);
            if (PERSIST) __syncthreads();   // the previous tile's last K-step may still be reading buffer 0/1
            sstore(0);
            __syncthreads();
            for (int kt = 0; kt < nk; ++kt) {
                const int buf = kt & 1;
                if (kt + 1 < nk) gload((kt + 1) * BK);
                mfma_tile_f32<BK, BM>(sA[buf], sB[buf], wm * 64 + li, wn * 64 + li, kh, acc);
                if (kt + 1 < nk) {
                    sstore(buf ^ 1);
                    __syncthreads();
                }
            }
        }
        const bool interior = (m0 + BM <= N1) && (n0 + BN <= N2);
        store_tile(out + (size_t)b * N1 * N2, acc, m0 + wm * 64, n0 + wn * 64, kh, li, N1, N2, interior);
    }
}

// ------------------------------------------------------------------------------------------------
// fp32, CHW operands, square volumes with N % 64 == 0: PERSISTENT kernel with a balanced mixed-tile schedule (round 2).
//
// What the one-tile-per-workgroup kernel above loses on the 640x480 shape (N = 4800, B = 2 pairs; measured):
//   * 2888 tiles of 128x128 on 256 CUs = 11.28 tiles per CU -> the last CU finishes 12: 6 % tail, plus 2.6 % dead rows /
//     columns in the 38th tile row / column (4800 = 37.5 x 128);
//   * it allocates 148 registers (84 VGPR, 32 of them hoisted LDS row addresses: the 8-bit offsets of ds_read2_b32 cannot
//     reach row k + 2 of a [BK][128] tile), i.e. 3 workgroups per CU, and runs FASTER with 2 (192 vs 201 us);
//   * the LDS image of the next K step is written in one burst of 16 ds_write_b128 per workgroup at the end of a stage,
//     which delays the other waves' fragment reads beyond their 256-cycle prefetch distance.
// This kernel: 2 workgroups per CU (512 on MI355X), each walking a STATIC schedule computed arithmetically from a few
// scalars (no table, no atomics): R_b rounds of 128x128 tiles covering the first n_big tiles of each pair in row-major
// order, then R_m rounds of 128x64 tiles and R_s rounds of 64x64 tiles covering exactly what is left (the rest of the
// last tile row, the remaining tile rows, the odd 64-wide column / row) — every workgroup does the same number of MFMA
// blocks to within one 64x64 tile, no dead rows.  For N = 4800, B = 2: 5 + 0 + 2 rounds (5 x 16 + 2 x 4 = 88 of the
// 45000 / 512 = 87.9 blocks per workgroup).  LDS tiles are stored column-permuted so that a wave's fragments of one k row
// are 256 B apart: every fragment read is ONE base register + immediates (ds_read2st64_b32), 126 registers in total; the
// register-staged next K step goes to LDS one float4 at a time between the MFMA groups.  The k loop order is unchanged
// (k ascending per output element): results are bitwise identical to the kernel above.
// Measured (profiles/probes/gemm3_probe.*): N = 4800, B = 2: 207.2 -> 190 us (72.4 -> 79 % of 157.3 TFLOP/s);
// steady state (B = 16): 78.3 -> 80.4 %, + interleaved stores 81.8 %.  Measured and NOT adopted: 256x128 / 256x256
// workgroup tiles (78.6 / 76.5 %), BK 8 / 32 / 64 (78.9 / 78.5 / 63 %), 1 / 3 / 4 workgroups per CU (74.6 / 80.5 / 77.8 %),
// fragment prefetch two groups ahead (no gain), the K stream running across tile boundaries (no cold prologue: no gain),
// de-phasing the two co-resident workgroups by a quarter tile (no gain).
// ------------------------------------------------------------------------------------------------
struct VolSched {   // host-computed, passed by value; everything per PAIR unless noted.  Cells are 64 x 64 outputs.
    int Nc, Gb;                     // cells per dimension; 128x128 tile grid per dimension (Nc / 2)
    int n_big_pp, full_rows, rem;   // big tiles of a pair: `full_rows` full tile rows + `rem` tiles of the next row
    int e, A, Bc, U_pp;             // 128x64 "units" not covered by big tiles: e per full row (A in total), Bc in the partial row
    int n_med_pp, n_small_pp;
    int R_b, R_m, R_s;              // rounds of each kind: every workgroup takes one item per round
    int B;
};

static bool make_vol_sched(int N, int B, int slots, VolSched& S) {
    if (N % 64 || N < 128 || B < 1 || slots < 8 || (slots & 7)) return false;
    S.B = B;
    S.Nc = N / 64;
    S.Gb = S.Nc / 2;
    const long cells = (long)B * S.Nc * S.Nc;
    long R_b = cells * 4 / 16 / slots;
    while (R_b > 0 && (R_b * slots % B != 0 || R_b * slots / B > (long)S.Gb * S.Gb)) --R_b;
    S.R_b = (int)R_b;
    S.n_big_pp = (int)(R_b * slots / B);
    S.full_rows = S.n_big_pp / S.Gb;
    S.rem = S.n_big_pp % S.Gb;
    S.e = S.Nc - 2 * S.Gb;
    S.A = S.full_rows * S.e;
    S.Bc = S.rem > 0 ? S.Nc - 2 * S.rem : 0;
    const int rows_after = S.Gb - S.full_rows - (S.rem > 0 ? 1 : 0);
    S.U_pp = S.A + S.Bc + rows_after * S.Nc;
    const long rem_cells = cells - 4L * S.n_big_pp * B;
    long R_m = rem_cells / 2 / slots;
    while (R_m > 0 && (R_m * slots % B != 0 || R_m * slots / B > S.U_pp)) --R_m;
    S.R_m = (int)R_m;
    S.n_med_pp = (int)(R_m * slots / B);
    S.n_small_pp = 2 * (S.U_pp - S.n_med_pp) + (S.e ? S.Nc : 0);
    S.R_s = (int)(((long)S.n_small_pp * B + slots - 1) / slots);
    // every cell covered exactly once, and the big tiles carry most of the work (otherwise the uniform kernel is better)
    return 4L * S.n_big_pp + 2L * S.n_med_pp + S.n_small_pp == (long)S.Nc * S.Nc && 4L * S.n_big_pp * 4 >= 3L * S.Nc * S.Nc;
}

// unit u of a pair -> (128-row tile row, 64-column cell)
__device__ __forceinline__ void vol_unit_coords(const VolSched& S, int u, int& tm, int& c) {
    if (u < S.A) { tm = u / S.e; c = 2 * S.Gb + (u - tm * S.e); return; }
    u -= S.A;
    if (u < S.Bc) { tm = S.full_rows; c = 2 * S.rem + u; return; }
    u -= S.Bc;
    const int r = u / S.Nc;
    tm = S.full_rows + (S.rem > 0 ? 1 : 0) + r;
    c = u - r * S.Nc;
}

// big tile q of a pair -> (tm, tn): super-rows of 16 tile rows walked column by column, so that the 64 consecutive items an
// XCD takes per round are a 16 x 4 block and its successive rounds continue to the right (16 row tiles = 2 MB stay in its
// L2 while 128-KB column tiles stream through once); the tiles of the partial row come last, in row order.
__device__ __forceinline__ void vol_big_coords(const VolSched& S, int q, int& tm, int& tn) {
    const int in_full = S.full_rows * S.Gb;
    if (q >= in_full) { tm = S.full_rows; tn = q - in_full; return; }
    constexpr int RG = 16;
    const int per_sr = RG * S.Gb;
    const int sr = q / per_sr, within = q - sr * per_sr;
    const int rows = min(RG, S.full_rows - sr * RG);
    tn = within / rows;
    tm = sr * RG + (within - tn * rows);
}

// One (64 MI) x (64 NJ) tile by the workgroup's 2 x 2 waves (wave tile 32 MI x 32 NJ), full K.  LDS: [stage][A: BK x 64 MI |
// B: BK x 64 NJ] floats, columns permuted (tile column w * 32 M + i * 32 + l -> i * 64 + w * 32 + l).
template <int MI, int NJ, class Pre>
__device__ __forceinline__ void vol_tile(const float* __restrict__ A, const float* __restrict__ Bp, float* __restrict__ O, int C,
                                         int N, int m0, int n0, float* smem, Pre&& pre_epilogue) {
    constexpr int BK = 16;
    constexpr int WA = 64 * MI, WB = 64 * NJ;             // operand tile widths (floats per k row)
    constexpr int TA = WA / 4, TB = WB / 4;               // loader threads per k row
    constexpr int NPA = BK * TA / 256, NPB = BK * TB / 256;   // float4 per thread per K step (2 or 1)
    constexpr int STAGE = BK * (WA + WB);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1, kh = lane >> 5, li = lane & 31;
    const int nk = C / BK, klast = C - BK;
    const int arow = t / TA, acol = (t % TA) * 4, brow = t / TB, bcol = (t % TB) * 4;
    // uniform (SGPR) tile bases + ONE 32-bit per-lane offset per operand (no 64-bit address VGPRs)
    const float* Au = A + m0;
    const float* Bu = Bp + n0;
    const unsigned aoff = arow * N + acol, boff = brow * N + bcol;
    auto perm = [](int col, int m) { return ((col >> 5) % m) * 64 + (col / (32 * m)) * 32 + (col & 31); };
    float* wa = smem + arow * WA + perm(acol, MI);
    float* wb = smem + BK * WA + brow * WB + perm(bcol, NJ);
    const float* fa = smem + kh * WA + wm * 32 + li;
    const float* fb = smem + BK * WA + kh * WB + wn * 32 + li;
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[2][NPA], rb[2][NPB];
    auto gload = [&](auto SET, int k0) {
        constexpr int S = decltype(SET)::value;
#pragma unroll
        for (int p = 0; p < NPA; ++p) ra[S][p] = *reinterpret_cast<const f32x4*>(Au + (size_t)(k0 + (256 / TA) * p) * N + aoff);
#pragma unroll
        for (int p = 0; p < NPB; ++p) rb[S][p] = *reinterpret_cast<const f32x4*>(Bu + (size_t)(k0 + (256 / TB) * p) * N + boff);
    };
    auto sstore = [&](auto SET, int buf) {
        constexpr int S = decltype(SET)::value;
#pragma unroll
        for (int p = 0; p < NPA; ++p) *reinterpret_cast<f32x4*>(wa + buf * STAGE + p * (256 / TA) * WA) = ra[S][p];
#pragma unroll
        for (int p = 0; p < NPB; ++p) *reinterpret_cast<f32x4*>(wb + buf * STAGE + p * (256 / TB) * WB) = rb[S][p];
    };
    // piece g (one float4) of the register-staged K step -> LDS; pieces 0..NPA-1 = A, then B
    auto spiece = [&](auto SET, int buf, int g) {
        constexpr int S = decltype(SET)::value;
#pragma unroll
        for (int p = 0; p < NPA; ++p)
            if (g == p) *reinterpret_cast<f32x4*>(wa + buf * STAGE + p * (256 / TA) * WA) = ra[S][p];
#pragma unroll
        for (int p = 0; p < NPB; ++p)
            if (g == NPA + p) *reinterpret_cast<f32x4*>(wb + buf * STAGE + p * (256 / TB) * WB) = rb[S][p];
    };
    // K step from LDS stage `buf`; fragments of k-pair kk + 2 are fetched before the MFMAs of k-pair kk are issued; after MFMA
    // group g one piece of the NEXT K step is stored (no store burst in front of the barrier)
    auto mma = [&](int buf, auto&& after) {
        const float* qa = fa + buf * STAGE;
        const float* qb = fb + buf * STAGE;
        float a[2][MI], b[2][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) a[0][i] = qa[i * 64];
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[0][j] = qb[j * 64];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
            if (kk + 2 < BK) {
#pragma unroll
                for (int i = 0; i < MI; ++i) a[nxt][i] = qa[(kk + 2) * WA + i * 64];
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[nxt][j] = qb[(kk + 2) * WB + j * 64];
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of the MFMAs (hipcc otherwise sinks it back)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            after(kk >> 1);
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    gload(S0{}, 0);
    gload(S1{}, BK);
    __syncthreads();   // the previous tile's last K step may still be reading the LDS stages
    sstore(S0{}, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {   // 2-deep register-staged global prefetch (loads stay in flight across the barrier)
        gload(S0{}, min((kt + 2) * BK, klast));
        mma(0, [&](int g) { spiece(S1{}, 1, g); });
        __syncthreads();
        gload(S1{}, min((kt + 3) * BK, klast));
        mma(1, [&](int g) { spiece(S0{}, 0, g); });
        __syncthreads();
    }
    // hook between the last K step and the epilogue stores: anything that must WAIT on a memory return belongs here (the
    // vmcnt queue is in order: behind the 64 stores below such a wait would sit until they have drained to memory)
    pre_epilogue();
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5): one full 128-B line per
    // half-wave and store instruction
    float* Ou = O + (size_t)m0 * N + n0;
    const unsigned so = (wm * 32 * MI + 4 * kh) * N + wn * 32 * NJ + li;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float* p = Ou + (size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * N;
#pragma unroll
            for (int j = 0; j < NJ; ++j) MV_VOL_STORE(acc[i][j][r], p + j * 32 + so);
        }
}

// The same tile with LDS-DMA staging (the default of the one-item-per-workgroup kernel): operand K steps go global -> LDS by
// global_load_lds_dwordx4 — no staging registers (96 instead of 116-126), no ds_write, and with nothing held in registers a K
// step can be 32 deep: NST LDS stages of BK k rows, the pieces of K step kt + NST - 1 issued between the MFMA groups of step kt,
// half as many barriers per tile with BK = 32.  The column permutation of the LDS image is applied to the SOURCE chunk each lane
// fetches (the DMA destination is lane-linear).  Waits are counted by hand: the only memory operations of a workgroup before
// its epilogue are its own DMA pieces, PPW per wave and K step, so "all but the newest PPW x (NST - 2)" = "K step kt has landed".
// k still ascends per output element: bitwise the register-staged kernel.
template <int MI, int NJ, int NST, int BK = 16>
__device__ __forceinline__ void vol_tile_dma(const float* __restrict__ A, const float* __restrict__ Bp, float* __restrict__ O, int C,
                                             int N, int m0, int n0, float* smem) {
    constexpr int WA = 64 * MI, WB = 64 * NJ;
    constexpr int STAGE = BK * (WA + WB);
    constexpr int PA = BK * WA / 256, PB = BK * WB / 256;   // 1-KB pieces per K step
    constexpr int PPW = (PA + PB) / 4;                      // per wave (4, 3 or 2)
    constexpr int RA = 256 / WA, RB = 256 / WB;             // k rows per piece
    constexpr int LA = NST - 1;                             // K steps a piece is issued ahead of its use
    const int t = threadIdx.x, lane = t & 63, wm = (t >> 6) >> 1, wn = (t >> 6) & 1, kh = lane >> 5, li = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nk = C / BK;
    const float* Au = A + m0;
    const float* Bu = Bp + n0;
    // lane -> (row inside the piece, position inside the LDS row) -> source column = inverse permutation of the position
    auto unperm = [](int pos, int m) { return ((pos & 63) >> 5) * 32 * m + (pos >> 6) * 32 + (pos & 31); };
    const unsigned voA = (unsigned)(((lane * 4) / WA) * N + unperm((lane * 4) % WA, MI)) * 4u;
    const unsigned voB = (unsigned)(((lane * 4) / WB) * N + unperm((lane * 4) % WB, NJ)) * 4u;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);
    auto piece = [&](int kt, int p) __attribute__((always_inline)) {   // piece p of this wave for K step kt -> stage kt % NST
        const int pi = wave + 4 * p;
        const bool isA = pi < PA;
        const int q = isA ? pi : pi - PA;
        const float* base = isA ? Au + (size_t)(kt * BK + q * RA) * N : Bu + (size_t)(kt * BK + q * RB) * N;
        const unsigned dst = lds0 + (unsigned)((kt % NST) * STAGE + (isA ? 0 : BK * WA) + q * 256) * 4u;
        glds16_s(isA ? voA : voB, base, dst);
    };
    const float* fa = smem + kh * WA + wm * 32 + li;
    const float* fb = smem + BK * WA + kh * WB + wn * 32 + li;
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto mma = [&](int buf, auto&& after) __attribute__((always_inline)) {
        const float* qa = fa + buf * STAGE;
        const float* qb = fb + buf * STAGE;
        float a[2][MI], b[2][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) a[0][i] = qa[i * 64];
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[0][j] = qb[j * 64];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
            if (kk + 2 < BK) {
#pragma unroll
                for (int i = 0; i < MI; ++i) a[nxt][i] = qa[(kk + 2) * WA + i * 64];
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[nxt][j] = qb[(kk + 2) * WB + j * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            after(kk >> 1);
        }
    };
#pragma unroll
    for (int k = 0; k < LA; ++k)                 // (nk >= LA is the launcher's condition)
#pragma unroll
        for (int p = 0; p < PPW; ++p) piece(k, p);
    int buf = 0;
    for (int kt = 0; kt <= nk - LA; ++kt) {
        wait_vmcnt_barrier<PPW * (LA - 1)>();     // K step kt has landed (LA - 1 later ones may be in flight); stage (kt + LA) % NST is free
        const bool more = kt + LA < nk;
        mma(buf, [&](int g) {
            if (g < PPW && more) piece(kt + LA, g);
        });
        buf = buf == NST - 1 ? 0 : buf + 1;
    }
    if constexpr (LA >= 3) {                     // tail: one K step fewer in flight each time
        wait_vmcnt_barrier<PPW * (LA - 2)>();
        mma(buf, [](int) {});
        buf = buf == NST - 1 ? 0 : buf + 1;
    }
    if constexpr (LA >= 2) {
        wait_vmcnt_barrier<0>();
        mma(buf, [](int) {});
    }
    float* Ou = O + (size_t)m0 * N + n0;
    const unsigned so = (wm * 32 * MI + 4 * kh) * N + wn * 32 * NJ + li;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float* p = Ou + (size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * N;
#pragma unroll
            for (int j = 0; j < NJ; ++j) MV_VOL_STORE(acc[i][j][r], p + j * 32 + so);
        }
}

// item (round r of the whole schedule, slot s inside the round) -> run the tile it denotes (or nothing).  XCD-major item
// numbering: XCD x (= s & 7: consecutive workgroup ids land on different XCDs) owns ONE contiguous run of each item list
// (R rounds x slots / 8 items), walked round by round — with the super-row order of vol_big_coords its big tiles form a compact
// 16-row x 20-column block of the output (N = 4800), so its 4 MB L2 sees 16 + 20 operand tiles instead of 8 x (8 + 8).
template <int DMA = 0, class Pre>
__device__ __forceinline__ void vol_run_item(const VolSched& S, int r, int s, int slots, const float* __restrict__ f1,
                                             const float* __restrict__ f2, float* __restrict__ out, int C, int N, size_t fsz,
                                             size_t osz, float* smem, Pre&& pre) {
    const int per = slots >> 3, xcd = s & 7, j = s >> 3;
    if (r < S.R_b) {
        const int idx = xcd * (S.R_b * per) + r * per + j;
        const int b = idx / S.n_big_pp;
        int tm, tn;
        vol_big_coords(S, idx - b * S.n_big_pp, tm, tn);
        if constexpr (DMA == 2) vol_tile_dma<2, 2, 2, 32>(f1 + b * fsz, f2 + b * fsz, out + b * osz, C, N, tm * 128, tn * 128, smem);
        else if constexpr (DMA > 0) vol_tile_dma<2, 2, DMA>(f1 + b * fsz, f2 + b * fsz, out + b * osz, C, N, tm * 128, tn * 128, smem);
        else vol_tile<2, 2>(f1 + b * fsz, f2 + b * fsz, out + b * osz, C, N, tm * 128, tn * 128, smem, pre);
        return;
    }
    r -= S.R_b;
    if (r < S.R_m) {
        const int idx = xcd * (S.R_m * per) + r * per + j;
        if (idx < S.n_med_pp * S.B) {
            const int b = idx / S.n_med_pp;
            int tm, c;
            vol_unit_coords(S, idx - b * S.n_med_pp, tm, c);
            if constexpr (DMA == 2) vol_tile_dma<2, 1, 2, 32>(f1 + b * fsz, f2 + b * fsz, out + b * osz, C, N, tm * 128, c * 64, smem);
            else if constexpr (DMA > 0) vol_tile_dma<2, 1, DMA>(f1 + b * fsz, f2 + b * fsz, out + b * osz, C, N, tm * 128, c * 64, smem);
            else vol_tile<2, 1>(f1 + b * fsz, f2 + b * fsz, out + b * osz, C, N, tm * 128, c * 64, smem, pre);
        } else {
            pre();
        }
        return;
    }
    r -= S.R_m;
    const int idx = xcd * (S.R_s * per) + r * per + j;
    if (idx < S.n_small_pp * S.B) {
        const int b = idx / S.n_small_pp, sm = idx - b * S.n_small_pp;
        const int from_units = 2 * (S.U_pp - S.n_med_pp);
        int m0, n0;
        if (sm < from_units) {   // the two 64x64 halves of a unit that did not become a 128x64 tile
            int tm, c;
            vol_unit_coords(S, S.n_med_pp + (sm >> 1), tm, c);
            m0 = tm * 128 + (sm & 1) * 64;
            n0 = c * 64;
        } else {                 // the odd last cell row
            m0 = (S.Nc - 1) * 64;
            n0 = (sm - from_units) * 64;
        }
        if constexpr (DMA == 2) vol_tile_dma<1, 1, 2, 32>(f1 + b * fsz, f2 + b * fsz, out + b * osz, C, N, m0, n0, smem);
        else if constexpr (DMA > 0) vol_tile_dma<1, 1, DMA>(f1 + b * fsz, f2 + b * fsz, out + b * osz, C, N, m0, n0, smem);
        else vol_tile<1, 1>(f1 + b * fsz, f2 + b * fsz, out + b * osz, C, N, m0, n0, smem, pre);
    } else {
        pre();
    }
}

// STATIC walk: workgroup s takes item s' of every round (s' = s with the XCD bits moved to the top, so that XCD x = s & 7 owns a
// contiguous eighth of each round).  Perfectly balanced when the kernel has the GPU to itself; used when no queue is available.
__global__ __launch_bounds__(256) void corr_volume_f32_sched(const float* __restrict__ f1, const float* __restrict__ f2,
                                                              float* __restrict__ out, int C, int N, VolSched S) {
    __shared__ __attribute__((aligned(16))) float smem[2 * 16 * 256];
    const int slots = gridDim.x;
    const int s = blockIdx.x;
    const size_t fsz = (size_t)C * N, osz = (size_t)N * N;
    const int rounds = S.R_b + S.R_m + S.R_s;
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) vol_run_item(S, r, s, slots, f1, f2, out, C, N, fsz, osz, smem, [] {});
}

// ONE ITEM PER WORKGROUP (default): grid = rounds x slots workgroups, workgroup i runs item (i / slots, i % slots) of the same
// schedule.  The hardware dispatcher hands workgroups out in id order as CU slots free up — big tiles first, 64x64 tiles last
// (longest processing time first) — so CUs that fall behind simply receive fewer items.  That matters inside the frame
// pipeline, where this GEMM shares the chip with the latency-bound kernels of the other streams: with the persistent static
// walk the slowest workgroup sets the kernel's end (measured: 190 us alone, 200 us beside one sequence's small kernels,
// 564 -> 896 us with three lanes).  Residency is held at 2 workgroups per CU (the measured optimum) by padding the
// workgroup's LDS request.  Also measured: a persistent grid pulling items from per-XCD atomic queues: 237 us — the returning
// atomic makes hipcc drain the vmcnt queue (the previous tile's 64 epilogue stores per wave) at every item, 4.5 us per tile.
__global__ __launch_bounds__(256) void corr_volume_f32_mixed(const float* __restrict__ f1, const float* __restrict__ f2,
                                                              float* __restrict__ out, int C, int N, VolSched S, int slots) {
    extern __shared__ __attribute__((aligned(16))) float smem_mixed[];   // 32 KB used + occupancy padding
    const int r = blockIdx.x / slots, s = blockIdx.x - r * slots;
    vol_run_item(S, r, s, slots, f1, f2, out, C, N, (size_t)C * N, (size_t)N * N, smem_mixed, [] {});
}

template <int NST>
__global__ __launch_bounds__(256) void corr_volume_f32_mixed_dma(const float* __restrict__ f1, const float* __restrict__ f2,
                                                                  float* __restrict__ out, int C, int N, VolSched S, int slots) {
    extern __shared__ __attribute__((aligned(16))) float smem_mixed_dma[];   // NST stages x 16 KB + occupancy padding
    const int r = blockIdx.x / slots, s = blockIdx.x - r * slots;
    vol_run_item<NST>(S, r, s, slots, f1, f2, out, C, N, (size_t)C * N, (size_t)N * N, smem_mixed_dma, [] {});
}

// ------------------------------------------------------------------------------------------------
// fp32, HWC ([N][C]) operands: same MFMA core; the loader transposes through LDS
// (global float4 along C -> 4 scalar LDS stores into the K-major tile; row padding +1 keeps them conflict-light).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void corr_volume_f32_hwc(const float* __restrict__ f1,
                                                            const float* __restrict__ f2,
                                                            float* __restrict__ out, int C, int N1, int N2,
                                                            int tiles_m, int tiles_n) {
    constexpr int BK = 16;
    constexpr int LD = BM + 4;  // padded leading dim (floats)
    __shared__ __attribute__((aligned(16))) float sA[2][BK][LD];
    __shared__ __attribute__((aligned(16))) float sB[2][BK][LD];

    int tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn);
    const int b = blockIdx.z;
    const int m0 = tm * BM, n0 = tn * BN;
    const float* A = f1 + (size_t)b * N1 * C;
    const float* Bp = f2 + (size_t)b * N2 * C;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // loader: tile = 128 rows x 16 k = 512 float4; thread t -> (row = t/4 + 64*p, k4 = (t%4)*4)
    const int lrow = t >> 2;
    const int lk = (t & 3) * 4;
    f32x4 ra[2], rb[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int ia = m0 + lrow + 64 * p, ib = n0 + lrow + 64 * p;
            ra[p] = (ia < N1) ? *reinterpret_cast<const f32x4*>(A + (size_t)ia * C + k0 + lk) : f32x4{0, 0, 0, 0};
            rb[p] = (ib < N2) ? *reinterpret_cast<const f32x4*>(Bp + (size_t)ib * C + k0 + lk) : f32x4{0, 0, 0, 0};
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sA[buf][lk + e][lrow + 64 * p] = ra[p][e];
                sB[buf][lk + e][lrow + 64 * p] = rb[p][e];
            }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    gload(0);
    sstore(0);
    __syncthreads();
    const int nk = C / BK;
    const int kh = lane >> 5, li = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
        mfma_tile_f32<BK, LD>(sA[buf], sB[buf], wm * 64 + li, wn * 64 + li, kh, acc);
        if (kt + 1 < nk) {
            sstore(buf ^ 1);
            __syncthreads();
        }
    }
    const bool interior = (m0 + BM <= N1) && (n0 + BN <= N2);
    store_tile(out + (size_t)b * N1 * N2, acc, m0 + wm * 64, n0 + wn * 64, kh, li, N1, N2, interior);
}


// ------------------------------------------------------------------------------------------------
// fp32 operands, HWC ([N][C]), bf16x3 SPLIT: each fp32 value is split on the fly into three bf16 pieces
// (x = hi + mid + lo, 8 + 8 + 8 significant bits, exact residuals) and the fp32 product a*b is rebuilt from the six
// bf16 MFMA products with i + j <= 2  (a0b2, a1b1, a2b0, a0b1, a1b0, a0b0 — smallest first), accumulated in fp32.
// Dropped terms are O(2^-24 |a||b|): fp32-class accuracy at 6/16 of the fp32-MFMA cycle cost (gfx950 has no
// TF32/xf32 MFMA; the reference itself runs this GEMM in TF32 or fp16, Module/Frontend/Frontend.py:275-278).
// LDS: six [128][32] bf16 tiles (3 pieces x 2 operands, 48 KB), 16-B chunks XOR-swizzled by (row>>2)&3.
// ------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));

// pre-pass: x[n] fp32 -> planes[3][n] bf16 (hi, mid, lo), x = hi + mid + lo up to 2^-24 |x|, residuals exact in fp32
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, uint16_t* __restrict__ planes, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        s16x4 h, m, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const __bf16 hb = (__bf16)v[e];
            const float r1 = v[e] - (float)hb;
            const __bf16 mb = (__bf16)r1;
            const float r2 = r1 - (float)mb;
            const __bf16 lb = (__bf16)r2;
            h[e] = __builtin_bit_cast(short, hb);
            m[e] = __builtin_bit_cast(short, mb);
            l[e] = __builtin_bit_cast(short, lb);
        }
        reinterpret_cast<s16x4*>(planes)[i] = h;
        reinterpret_cast<s16x4*>(planes + n4 * 4)[i] = m;
        reinterpret_cast<s16x4*>(planes + 2 * n4 * 4)[i] = l;
    }
}

// GEMM over pre-split planes: operands are [3][B][N][C] bf16 (piece-major).  Per K-tile of 32 the six [128][32] bf16
// tiles (3 pieces x 2 operands, 48 KB, single buffer, register prefetch of the next tile) feed 6 x 4 x 2 = 48 MFMAs per
// wave; 3 workgroups per CU interleave their load / MFMA phases.  No VALU work besides addressing.
// NP = 3: all six products (fp32-class).  NP = 2: only the two leading pieces of each operand (16 significant bits, finer
// than TF32's 11) and the three products a0b1, a1b0, a0b0 — the precision class the reference's own fast frontend runs
// this GEMM in (allow_tf32 / matmul precision "medium", Module/Frontend/Frontend.py:275-277); 32 KB LDS, 4 workgroups / CU.
template <int NP>
__global__ __launch_bounds__(256) void corr_volume_bf16x3_hwc(const uint16_t* __restrict__ p1,
                                                               const uint16_t* __restrict__ p2,
                                                               float* __restrict__ out, int C, int N1, int N2, int Bn,
                                                               int tiles_m, int tiles_n) {
    constexpr int BK = 32;
    constexpr int CH = BK / 8;                   // 16-B chunks per row = 4
    constexpr int TILE = BM * CH;                // s16x8 elements per [128][32] tile
    __shared__ __attribute__((aligned(16))) s16x8 smem[2 * NP * TILE];

    int tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn);
    const int b = blockIdx.z;
    const int m0 = tm * BM, n0 = tn * BN;
    const size_t plane1 = (size_t)Bn * N1 * C, plane2 = (size_t)Bn * N2 * C;
    const uint16_t* A = p1 + (size_t)b * N1 * C;
    const uint16_t* Bp = p2 + (size_t)b * N2 * C;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int kh = lane >> 5, li = lane & 31;

    // loader: per tile 128 rows x 4 chunks = 512 chunks; thread t -> (row = t/4 + 64*p, chunk = t%4), p < 2
    const int lrow = t >> 2, lch = t & 3;
    s16x8 r[2 * NP][2];
    const s16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    auto gload = [&](int k0) {
#pragma unroll
        for (int pc = 0; pc < NP; ++pc)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int ia = m0 + lrow + 64 * p, ib = n0 + lrow + 64 * p;
                r[pc][p] = (ia < N1) ? *reinterpret_cast<const s16x8*>(A + pc * plane1 + (size_t)ia * C + k0 + lch * 8) : zero;
                r[NP + pc][p] = (ib < N2) ? *reinterpret_cast<const s16x8*>(Bp + pc * plane2 + (size_t)ib * C + k0 + lch * 8) : zero;
            }
    };
    auto swz = [](int row, int c) { return row * CH + (c ^ ((row >> 2) & 3)); };
    auto sstore = [&]() {
#pragma unroll
        for (int tl = 0; tl < 2 * NP; ++tl)
#pragma unroll
            for (int p = 0; p < 2; ++p) smem[tl * TILE + swz(lrow + 64 * p, lch)] = r[tl][p];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    const int nk = C / BK;
    gload(0);
    for (int kt = 0; kt < nk; ++kt) {
        sstore();
        __syncthreads();
        if (kt + 1 < nk) gload((kt + 1) * BK);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int c = ks * 2 + kh;
            bf16x8 a[NP][2], bb[NP][2];
#pragma unroll
            for (int pc = 0; pc < NP; ++pc)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[pc][i] = __builtin_bit_cast(bf16x8, smem[pc * TILE + swz(wm * 64 + i * 32 + li, c)]);
                    bb[pc][i] = __builtin_bit_cast(bf16x8, smem[(NP + pc) * TILE + swz(wn * 64 + i * 32 + li, c)]);
                }
            // smallest products first: (a0,b2) (a1,b1) (a2,b0) (a0,b1) (a1,b0) (a0,b0)
            constexpr int NQ = NP == 3 ? 6 : 3;
            constexpr int PA[6] = {NP == 3 ? 0 : 0, 1, NP == 3 ? 2 : 0, 0, 1, 0};
            constexpr int PB[6] = {NP == 3 ? 2 : 1, NP == 3 ? 1 : 0, 0, 1, 0, 0};
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]][i], bb[PB[q]][j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    const bool interior = (m0 + BM <= N1) && (n0 + BN <= N2);
    store_tile(out + (size_t)b * N1 * N2, acc, m0 + wm * 64, n0 + wn * 64, kh, li, N1, N2, interior);
}

// ------------------------------------------------------------------------------------------------
// 16-bit operands (f16 / bf16), HWC ([N][C]): K is contiguous, which is the 32x32x16 fragment order
// (lane l holds 8 consecutive k of row l&31, k-group l>>5).  BK = 64: a tile row is one 128-B line.
// LDS tile [128 rows][64 k] 16-bit, 16-B chunks XOR-swizzled by (row & 7) so ds_read_b128 is conflict-free.
// ------------------------------------------------------------------------------------------------
template <bool IS_BF16, int BK, bool NT = true>
__global__ __launch_bounds__(256) void corr_volume_h_hwc(const uint16_t* __restrict__ f1,
                                                          const uint16_t* __restrict__ f2,
                                                          float* __restrict__ out, int C, int N1, int N2,
                                                          int tiles_m, int tiles_n) {
    // BK 16-bit elements per tile row: 64 -> one 128-B line per row, 64 KB LDS, 2 workgroups / CU;
    //                                  32 -> half lines, 32 KB LDS, 4 workgroups / CU (more loads in flight: the kernel
    //                                  is latency-bound, its MFMA work is ~1 us per workgroup)
    constexpr int CH = BK / 8;              // 16-byte chunks per row
    constexpr int RPP = 256 / CH;           // rows covered per loader pass
    constexpr int NPASS = BM / RPP;
    __shared__ __attribute__((aligned(16))) s16x8 sA[2][BM * CH];
    __shared__ __attribute__((aligned(16))) s16x8 sB[2][BN * CH];

    int tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn);
    const int b = blockIdx.z;
    const int m0 = tm * BM, n0 = tn * BN;
    const uint16_t* A = f1 + (size_t)b * N1 * C;
    const uint16_t* Bp = f2 + (size_t)b * N2 * C;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int lrow = t / CH, lch = t % CH;
    s16x8 ra[NPASS], rb[NPASS];
    const s16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    auto gload = [&](int k0) {
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int ia = m0 + lrow + RPP * p, ib = n0 + lrow + RPP * p;
            ra[p] = (ia < N1) ? *reinterpret_cast<const s16x8*>(A + (size_t)ia * C + k0 + lch * 8) : zero;
            rb[p] = (ib < N2) ? *reinterpret_cast<const s16x8*>(Bp + (size_t)ib * C + k0 + lch * 8) : zero;
        }
    };
    // conflict-free ds_read_b128: a 16-lane service group must hit 16 distinct 16-B slots of the 256-B bank row
    auto swz = [](int row, int c) { return row * CH + (c ^ (CH == 8 ? (row & 7) : ((row >> 2) & 3))); };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int row = lrow + RPP * p;
            sA[buf][swz(row, lch)] = ra[p];
            sB[buf][swz(row, lch)] = rb[p];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    gload(0);
    sstore(0);
    __syncthreads();
    const int nk = C / BK;
    const int kh = lane >> 5, li = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {   // MFMA k-steps of 16
            const int ch = ks * 2 + kh;          // 16-B chunk holding this lane's 8 k values
            s16x8 a[2], bb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = sA[buf][swz(wm * 64 + i * 32 + li, ch)];
#pragma unroll
            for (int j = 0; j < 2; ++j) bb[j] = sB[buf][swz(wn * 64 + j * 32 + li, ch)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (IS_BF16)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, bb[j]), acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            __builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, bb[j]), acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) {
            sstore(buf ^ 1);
            __syncthreads();
        }
    }
    const bool interior = (m0 + BM <= N1) && (n0 + BN <= N2);
    store_tile<NT>(out + (size_t)b * N1 * N2, acc, m0 + wm * 64, n0 + wn * 64, kh, li, N1, N2, interior);
}

// ------------------------------------------------------------------------------------------------
// 16-bit operands, HWC, C = 128 / 256: STREAMING form (round 2).  With 16-bit inputs the volume is bound by its 4-byte
// output (184 MB at 640x480 against 9.4 us of MFMA work), so the kernel is shaped like a fill kernel with a GEMM attached:
//   * persistent 256-thread workgroups (two per CU) each own a contiguous run of (128-row band, 64-column sub-tile) items;
//     a wave keeps the whole-K A fragments of its 32 rows in REGISTERS for a band (KS x 4 VGPRs) and only 64-column B
//     sub-tiles (64 x C x 2 B) stream through a 2-slot LDS ring: ~1.1 B loaded per B stored instead of 2.0 for 128x128 tiles
//     rebuilt per K step, and half the LDS reads;
//   * the stores of sub-tile j are interleaved with the MFMAs of sub-tile j + 1 (two accumulator sets): in-kernel cycle
//     stamps (profiles/probes/hstream_probe.*) showed the un-pipelined form spending 1800 of 4800 cycles per sub-tile just
//     ISSUING its 32 stores against the write path's back-pressure, with the memory system idle during the other phases;
//   * one barrier per sub-tile, loads of sub-tile j + 2 in flight while j is multiplied.  The B loads are inline-asm
//     global_load_dwordx4 with hand-placed s_waitcnt vmcnt(32): hipcc's own wait insertion merges the memory state of every
//     path into a loop, and any conditional load or store there degrades all waits to vmcnt(0) — which drains the 32
//     stores in flight each sub-tile (measured: 65 us).  vmcnt is an in-order counter, so "at most 32 outstanding" means "the
//     loads issued before the last 32 stores have landed";
//   * a wave owns 32 rows x 64 columns of the sub-tile: 2 x KS v_mfma_f32_32x32x16 per sub-tile, B fragments by ds_read_b128
//     fetched three k-steps ahead from XOR-swizzled rows (chunk ^ (row & 15): conflict-free for the hardware's 16-lane
//     service groups), full-line stores.
// Items are split evenly over the grid (ceil(T / G) vs floor: one sub-tile of imbalance), no tail.
// ------------------------------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));

// OUT16 (round 4, the Fast-mode volume AS THE REFERENCE COMPUTES IT): with the encoder in fp16 / bf16 (MACVO_Fast.yaml:73-74) `einsum` returns the
// volume in that 16-bit type and flownet.py:27 merely widens it.  The epilogue then rounds the fp32 accumulators ONCE (round-to-nearest-even,
// what the library GEMM's epilogue does) and stores 2-byte cells: 92 MB per 640x480 frame instead of 184 MB for a kernel that is bound by its
// output.  The MFMA C layout gives a lane ONE column per accumulator set; here the two sets of a wave take the EVEN and the ODD columns of the
// 64-column sub-tile (B fragment rows 2 li and 2 li + 1 instead of li and li + 32; the ring's XOR swizzle keys on row / 2 so that the reads stay
// conflict-free), so a lane holds columns (2 li, 2 li + 1) of its 16 rows: one v_cvt_pk + ONE dword store per row, 32 lanes = one full 128-byte
// line, 16 stores per item instead of 32 (the hand-counted vmcnt constants follow: NST).  (First form, measured and dropped: columns li / li + 32,
// row pairs swapped between neighbouring lanes by DPP, 64-byte half-line nt stores — 63 us against 40 us for the fp32 kernel.)
#ifdef MV_HS_STAMPS   // probe builds only: s_memtime stamps of the first 48 items of every wave of the first 32 workgroups -> [wg][wave][item][4]
__device__ long long* g_hs_stamps = nullptr;
#define HS_STAMP(i) do { if (stamp_on) st[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HS_STAMP(i) ((void)0)
#endif
template <bool IS_BF16, int KS, bool OUT16 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void corr_volume_h_stream(
    const uint16_t* __restrict__ f1, const uint16_t* __restrict__ f2, float* __restrict__ out, int N1, int N2, int B, int R) {
    constexpr int C = KS * 16, CH = KS * 2;      // channels; 16-byte chunks per row
    constexpr int NST = OUT16 ? 16 : 32;         // stores per wave and item
    constexpr int OSZ = OUT16 ? 2 : 4;           // bytes per output cell
    constexpr int RPI = 64 / CH;                 // B rows one wave-wide LDS-DMA instruction covers (64 lanes x 16 B = 1 KB)
    constexpr int NP = 64 / (4 * RPI);           // LDS-DMA instructions per wave and sub-tile (4 waves)
    constexpr int RPK = 16 / KS;                 // accumulator rows stored per k-step of the NEXT sub-tile
    constexpr int SLOT = 64 * CH;                // ring slot in 16-byte units
    static_assert(NP <= 8 && RPK * KS == 16, "the vmcnt budget below assumes <= 8 loads and 32 stores per sub-tile");
    extern __shared__ __attribute__((aligned(16))) i32x4 smem_hs[];   // B ring: 2 slots x 64 rows x CH chunks
#ifdef MV_HS_STAMPS
    const bool stamp_on = g_hs_stamps != nullptr && blockIdx.x < 32;
    long long st[4] = {0, 0, 0, 0};
    int n_st = 0;
#endif
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int kh = lane >> 5, li = lane & 31;
    const int nb = (N1 + 127) >> 7, nc = N2 >> 6;
    // Item order: pair, then one of R column REGIONS (sub-tiles [g * nc / R, (g + 1) * nc / R)), then band, then sub-tile inside the
    // region.  The list is cut into 8 equal runs, one per XCD (workgroup id % 8 = XCD under round-robin dispatch), and each
    // XCD's run into equal runs for its gridDim / 8 workgroups: what an XCD reads at any time is ONE region's B rows (host picks
    // R so that this is ~1.2 MB, re-read once per band) plus the bands its workgroups are on — resident in its 4 MB L2 while
    // 23 MB of output stream through it.  Measured (profiles/probes/store_probe.py): L2-hitting reads beside the 184 MB write stream
    // are free, L2-missing ones cost ~7 us per 45 MB — with the plain band-major split the kernel fetched 66-132 MB per launch.
    const int per = nb * nc, T = B * per;        // T < 2^31
    int it, it_end;
    {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3, nj = gridDim.x >> 3;
        const long lo = (long)x * T / 8, hi = ((long)x + 1) * T / 8;
        it = (int)(lo + (hi - lo) * j / nj);
        it_end = (int)(lo + (hi - lo) * (j + 1) / nj);
    }
    if (it >= it_end) return;
    auto reg_c0 = [&](int g) { return (int)((long)g * nc / R); };       // first sub-tile of region g
    auto decode = [&](int i, int& b, int& g, int& band, int& c) {       // rare: once per run and per segment
        b = i / per;
        int rem = i - b * per;
        g = 0;
        while (g + 1 < R && rem >= nb * reg_c0(g + 1)) ++g;
        rem -= nb * reg_c0(g);
        const int w = reg_c0(g + 1) - reg_c0(g);
        band = rem / w;
        c = reg_c0(g) + (rem - band * w);
        b = __builtin_amdgcn_readfirstlane(b);   // (they are uniform; this lets hipcc keep everything derived from them in SGPRs)
        g = __builtin_amdgcn_readfirstlane(g);
        band = __builtin_amdgcn_readfirstlane(band);
        c = __builtin_amdgcn_readfirstlane(c);
    };
    s16x8 af[KS];
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem_hs);   // low 32 bits of a flat LDS pointer = LDS byte offset
    // ---- B loader: LDS-DMA, one item ahead of the MFMAs, all walking state wave-uniform ----
    // ring slot layout: row-major [64][CH] chunks, chunk c of row r stored at position c ^ (r & 15) — the DMA destination is
    // lane-linear, so the permutation is applied to the SOURCE chunk each lane fetches (same involution as the fragment reads)
    int ld_it = it, ld_b, ld_g, ld_band, ld_c, ld_c0, ld_cend;
    decode(it, ld_b, ld_g, ld_band, ld_c);
    ld_c0 = reg_c0(ld_g);
    ld_cend = reg_c0(ld_g + 1);
    const uint16_t* ld_ptr = f2 + ((size_t)ld_b * N2 + (size_t)ld_c * 64) * C;   // f2 row 0 of the (pair, sub-tile) of item ld_it
    unsigned lane_src[NP];                       // element offset of this lane's source chunk inside the sub-tile, per instruction
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int row = (p * 4 + wave) * RPI + lane / CH, pos = lane % CH;
        lane_src[p] = (unsigned)(row * C + ((pos ^ ((OUT16 ? row >> 1 : row) & 15)) << 3));
    }
    auto issue_piece = [&](int slot, int p) __attribute__((always_inline)) {   // piece p (constant after unrolling) of item ld_it -> ring slot
#ifndef MV_HS_PROBE_NODMA                        // (probe builds: timing only)
        glds16(ld_ptr + lane_src[p], lds0 + (unsigned)(slot * SLOT + (p * 4 + wave) * 64) * 16u);
#endif
    };
    auto advance_b = [&]() __attribute__((always_inline)) {          // behind the last piece of an item
        if (ld_it + 1 < it_end) {                // past the end of the run the last item is simply fetched again
            ++ld_it;
            if (++ld_c == ld_cend) {             // next band of the region / next region / next pair
                if (++ld_band == nb) {
                    ld_band = 0;
                    if (++ld_g == R) {
                        ld_g = 0;
                        ++ld_b;
                    }
                    ld_c0 = reg_c0(ld_g);
                    ld_cend = reg_c0(ld_g + 1);
                }
                ld_c = ld_c0;
            }
            ld_ptr = f2 + ((size_t)ld_b * N2 + (size_t)ld_c * 64) * C;
        }
    };
    auto issue_b = [&](int slot) __attribute__((always_inline)) {   // prologue form: the NP pieces as one block
#pragma unroll
        for (int p = 0; p < NP; ++p) issue_piece(slot, p);
        advance_b();
    };
    issue_b(0);
    int slot = 0;
    const int sw = li & 15;                      // fragment rows li and li + 32 share the swizzle
    float* O = nullptr;                          // wave-uniform: column 0 of the CURRENT item's output block (row 0 of the pair)
    constexpr int OADV = OUT16 ? 32 : 64;        // one item = 64 columns, in units of sizeof(float)
    // per-lane BYTE offsets of the 16 accumulator rows (C/D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)),
    // fixed for a band: stores are `uniform base + 32-bit lane offset`, no address arithmetic between the MFMAs.
    // Rows past N1 (last band) hold copies of row N1 - 1 (the A rows are clamped the same way): they are stored ON TOP of row
    // N1 - 1 with identical values instead of being branched around.
    unsigned roff[16];
    auto pack16 = [&](float lo, float hi) __attribute__((always_inline)) -> unsigned {
        if (IS_BF16) return (unsigned)__builtin_bit_cast(uint16_t, (__bf16)lo) | ((unsigned)__builtin_bit_cast(uint16_t, (__bf16)hi) << 16);
        return (unsigned)__builtin_bit_cast(uint16_t, (_Float16)lo) | ((unsigned)__builtin_bit_cast(uint16_t, (_Float16)hi) << 16);
    };
    auto store_r = [&](const f32x16& p0, const f32x16& p1, int r, float* Ob) __attribute__((always_inline)) {
        // asm: hipcc strength-reduces `Ob + roff[r]` into sixteen 64-bit per-lane pointers (32 VGPRs -> spills at the 256-register
        // budget of 2 waves / SIMD); the SGPR-base form needs none.  Like the DMA above these are counted by hand.
        if (OUT16) {                             // columns (2 li, 2 li + 1) of row r: one packed dword, a full line per 32 lanes
            const unsigned d = pack16(p0[r], p1[r]);
#ifdef MV_HS_PROBE_NOSTORE                       // (probe builds: timing only)
            asm volatile("" ::"v"(d), "v"(roff[r]), "s"(Ob));
#else
            asm volatile("global_store_dword %0, %1, %2" MV_VOL_STORE_ASM_MOD ::"v"(roff[r]), "v"(d), "s"(Ob) : "memory");
#endif
        } else {
            asm volatile("global_store_dword %0, %1, %2" MV_VOL_STORE_ASM_MOD ::"v"(roff[r]), "v"(p0[r]), "s"(Ob) : "memory");
            asm volatile("global_store_dword %0, %1, %2 offset:128" MV_VOL_STORE_ASM_MOD ::"v"(roff[r]), "v"(p1[r]), "s"(Ob) : "memory");
        }
    };
    // one sub-tile: ring upkeep, 2 x KS MFMAs into (c0, c1); with PREV the 32 stores of (p0, p1) ride between them
    auto step = [&](auto PREV, f32x16& c0, f32x16& c1, const f32x16& p0, const f32x16& p1) __attribute__((always_inline)) {
        constexpr bool HAVE_PREV = decltype(PREV)::value;
        HS_STAMP(0);
        // this wave's DMA pieces of ring slot `slot` were issued before the last 32 stores (or everything has been drained since);
        // behind the barrier all four waves' pieces are in, and everyone has left slot ^ 1
#ifndef MV_HS_PROBE_SLACK
#define MV_HS_PROBE_SLACK 0       // probe builds only (timing; results are then wrong): that many more stores may be in flight at the barrier
#endif
        wait_vmcnt_barrier<NST + MV_HS_PROBE_SLACK>();
        HS_STAMP(1);
        // Round 6: the NP LDS-DMA pieces of the NEXT item are no longer issued as a block in front of the MFMAs (in-kernel stamps, profiles/r06_hs_stamps.log: the
        // block took 0.62 of an item's 3.1 us — the wave's VMEM queue is still full of the previous item's stores — with the matrix pipe idle) but DPK per k-step behind
        // the MFMAs of the first SH k-steps; the stores of the previous item follow in the remaining k-steps (twice as many per k-step in the last SH).  The ORDER of this
        // wave's memory operations is unchanged (all pieces, then all stores), so the hand-counted vmcnt arithmetic is too.
        constexpr int SH = KS / 4, DPK = NP / SH;
        static_assert(NP % SH == 0 && SH >= 1, "pieces per k-step");
        HS_STAMP(2);
#pragma unroll
        for (int r = 0; r < 16; ++r) c0[r] = c1[r] = 0.f;
        const i32x4* q0 = smem_hs + slot * SLOT + (OUT16 ? 2 * li : li) * CH;      // OUT16: even / odd columns (rows 2 li, 2 li + 1 of the B sub-tile)
        const i32x4* q1 = q0 + (OUT16 ? 1 : 32) * CH;
        constexpr int PF = 2;                    // B fragments are fetched PF k-steps ahead of the MFMAs that use them
        i32x4 fb0[PF + 1], fb1[PF + 1];
#pragma unroll
        for (int ks = 0; ks < PF; ++ks) {
            fb0[ks] = q0[(ks * 2 + kh) ^ sw];
            fb1[ks] = q1[(ks * 2 + kh) ^ sw];
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#ifndef MV_HS_PROBE_NOLDS
            if (ks + PF < KS) {
                fb0[(ks + PF) % (PF + 1)] = q0[((ks + PF) * 2 + kh) ^ sw];
                fb1[(ks + PF) % (PF + 1)] = q1[((ks + PF) * 2 + kh) ^ sw];
            }
#endif
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of the MFMAs (hipcc otherwise sinks it back)
            const i32x4 b0 = fb0[ks % (PF + 1)], b1 = fb1[ks % (PF + 1)];
#ifdef MV_HS_PROBE_NOMFMA
            asm volatile("" ::"v"(b0), "v"(b1), "v"(af[ks]));
            if (false) {
#else
            if (IS_BF16) {
#endif
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[ks]), __builtin_bit_cast(bf16x8, b0), c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[ks]), __builtin_bit_cast(bf16x8, b1), c1, 0, 0, 0);
            } else {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[ks]), __builtin_bit_cast(f16x8, b0), c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[ks]), __builtin_bit_cast(f16x8, b1), c1, 0, 0, 0);
            }
            if (ks < SH) {
#pragma unroll
                for (int q = 0; q < DPK; ++q) issue_piece(slot ^ 1, ks * DPK + q);
                if (ks == SH - 1) advance_b();
            } else if (HAVE_PREV) {
                const int first = (ks - SH) * RPK + (ks > KS - SH ? (ks - (KS - SH)) * RPK : 0);
                const int count = ks >= KS - SH ? 2 * RPK : RPK;
#pragma unroll
                for (int q = 0; q < count; ++q) store_r(p0, p1, first + q, O - OADV);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!HAVE_PREV) wait_vmcnt<0>();         // no stores went out behind this pass's DMA: the next pass's vmcnt(NST) would not cover it
        HS_STAMP(3);
#ifdef MV_HS_STAMPS
        if (stamp_on && (threadIdx.x & 63) == 0 && n_st < 48) {
            long long* o = g_hs_stamps + (((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 48 + n_st) * 4;
            o[0] = st[0]; o[1] = st[1]; o[2] = st[2]; o[3] = st[3];
        }
        ++n_st;
#endif
        ++it;
        slot ^= 1;
        O += OADV;
    };
    auto flush = [&](const f32x16& p0, const f32x16& p1) __attribute__((always_inline)) {
        // asm stores straight behind the last MFMAs: "XDL write VGPR -> VMEM read" is a software hazard (11 wait states for an
        // 8-pass MFMA) that hipcc's hazard recognizer does not see through inline asm
        asm volatile("s_nop 15" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r) store_r(p0, p1, r, O - OADV);
    };
    using Yes = std::true_type;
    using No = std::false_type;
    // whole-K A fragments of this wave's 32 rows of (pair b, band) -> registers.  asm like the DMA: the wait is placed by hand so
    // that the previous band's last 32 stores can be issued BEHIND these loads and drain while the first sub-tile is multiplied
    // (a compiler-tracked load would be waited for with vmcnt(0), i.e. together with those stores: vmcnt retires in order).
    // Caveat: to hipcc an asm load's destination is defined at the end of the statement; under register pressure it may copy or
    // spill such a value before the hand-placed wait (this happened in the fp32 streaming kernel, which therefore uses tracked
    // buffer loads).  Here the values stay put (no scratch, the "+v" re-definition below sits behind the wait), and
    // test_corr_volume_16bit_streaming_form_is_bitwise_the_tile_form runs multi-band workgroups, so a build where they do not
    // fails deterministically.
    i32x4 afr[KS];
    auto issue_a = [&](int b, int band) __attribute__((always_inline)) {
        const int gr = min(band * 128 + wave * 32 + li, N1 - 1);
        const uint16_t* A = f1 + ((size_t)b * N1 + gr) * C + kh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(afr[ks]) : "v"(A), "n"(ks * 32) : "memory");
    };
    f32x16 x0, x1, y0, y1;
    int b, g, band, c0i;
    decode(it, b, g, band, c0i);
    issue_a(b, band);
    wait_vmcnt<0>();
    while (true) {                               // one pass per (pair, region, band) segment of the run
        const int seg_end = min(it_end, it + (reg_c0(g + 1) - c0i));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            asm volatile("" : "+v"(afr[ks]));    // the fragments count as defined only here, behind the hand-placed wait
            af[ks] = __builtin_bit_cast(s16x8, afr[ks]);
        }
        O = reinterpret_cast<float*>(reinterpret_cast<char*>(out) + ((size_t)b * N1 * N2 + (size_t)c0i * 64) * OSZ);
#pragma unroll
        for (int r = 0; r < 16; ++r)             // (OUT16: the lane's dword holds columns 2 li, 2 li + 1)
            roff[r] = ((unsigned)min(band * 128 + wave * 32 + 4 * kh + (r & 3) + 8 * (r >> 2), N1 - 1) * (unsigned)N2 + (OUT16 ? 2 * li : li)) * (unsigned)OSZ;
        step(No{}, x0, x1, x0, x1);
        while (it + 2 <= seg_end) {
            step(Yes{}, y0, y1, x0, x1);
            step(Yes{}, x0, x1, y0, y1);
        }
        const bool odd = it < seg_end;
        if (odd) step(Yes{}, y0, y1, x0, x1);
        const bool more = it < it_end;           // (uniform) next segment: its A loads go out first, the flush rides behind them
        if (more) {
            decode(it, b, g, band, c0i);
            issue_a(b, band);
        }
        if (odd) flush(y0, y1);
        else flush(x0, x1);
        if (!more) break;
        wait_vmcnt<NST>();                       // the A loads precede the NST flush stores
    }
}

// (round 2: an fp32 STREAMING form of this GEMM — persistent workgroups, whole-K A fragments in registers, B through an LDS-DMA ring in K halves —
// was built, gave the tile kernels' bits and ran 193 us against their 186: measured and not adopted, removed in round 5.  DESIGN.md changelog.)

// ------------------------------------------------------------------------------------------------
// 16-bit operands, CHW ([C][N]): K-major.  The tile is staged K-major in LDS ([BK][128]) and each lane
// assembles its 8-k fragment with 8 ds_read_u16 (column li, rows 8*kh .. 8*kh+7 of the k-step).
// Compute is ~6% of the kernel at this size (HBM-write bound), so the scalar fragment gather is
// acceptable for the layout the reference hands over; HWC is the fast path for 16-bit features.
// ------------------------------------------------------------------------------------------------
template <bool IS_BF16>
__global__ __launch_bounds__(256) void corr_volume_h_chw(const uint16_t* __restrict__ f1,
                                                          const uint16_t* __restrict__ f2,
                                                          float* __restrict__ out, int C, int N1, int N2,
                                                          int tiles_m, int tiles_n) {
    constexpr int BK = 32;
    constexpr int LD = BM + 8;  // 16-bit elements; +8 keeps 16-B alignment of row starts and skews banks
    __shared__ __attribute__((aligned(16))) uint16_t sA[2][BK][LD];
    __shared__ __attribute__((aligned(16))) uint16_t sB[2][BK][LD];

    int tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn);
    const int b = blockIdx.z;
    const int m0 = tm * BM, n0 = tn * BN;
    const uint16_t* A = f1 + (size_t)b * C * N1;
    const uint16_t* Bp = f2 + (size_t)b * C * N2;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // loader: 32 rows x 128 cols = 512 chunks of 8; thread t -> (row = t/16 + 16*p, col8 = (t%16)*8), p < 2
    const int lrow = t >> 4, lcol = (t & 15) * 8;
    const bool vec_ok = ((N1 & 7) == 0) && ((N2 & 7) == 0);
    s16x8 ra[2], rb[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int k = k0 + lrow + 16 * p;
            const uint16_t* pa = A + (size_t)k * N1 + m0 + lcol;
            const uint16_t* pb = Bp + (size_t)k * N2 + n0 + lcol;
            if (vec_ok) {
                const s16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
                ra[p] = (m0 + lcol < N1) ? *reinterpret_cast<const s16x8*>(pa) : zero;
                rb[p] = (n0 + lcol < N2) ? *reinterpret_cast<const s16x8*>(pb) : zero;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    ra[p][e] = (m0 + lcol + e < N1) ? (short)pa[e] : (short)0;
                    rb[p][e] = (n0 + lcol + e < N2) ? (short)pb[e] : (short)0;
                }
            }
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            *reinterpret_cast<s16x8*>(&sA[buf][lrow + 16 * p][lcol]) = ra[p];
            *reinterpret_cast<s16x8*>(&sB[buf][lrow + 16 * p][lcol]) = rb[p];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    gload(0);
    sstore(0);
    __syncthreads();
    const int nk = C / BK;
    const int kh = lane >> 5, li = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int kb = ks * 16 + kh * 8;
            s16x8 a[2], bb[2];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i][e] = (short)sA[buf][kb + e][wm * 64 + i * 32 + li];
#pragma unroll
                for (int j = 0; j < 2; ++j) bb[j][e] = (short)sB[buf][kb + e][wn * 64 + j * 32 + li];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (IS_BF16)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, bb[j]), acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            __builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, bb[j]), acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) {
            sstore(buf ^ 1);
            __syncthreads();
        }
    }
    const bool interior = (m0 + BM <= N1) && (n0 + BN <= N2);
    store_tile(out + (size_t)b * N1 * N2, acc, m0 + wm * 64, n0 + wn * 64, kh, li, N1, N2, interior);
}

}  // namespace

// Persistent-grid size of the fp32 GEMM: 3 workgroups per CU (queried once), rounded down to a multiple of 8 (XCDs).
// Opt-in: MV_VOL_PERSIST_WG_PER_CU=<k> (default 0 = classic one-tile-per-workgroup grid, which measured the same GEMM
// time; on the round-1 test system the expected cross-stream overlap did not materialise either way — lookups launched
// beside the GEMM ran ~3x slower and finished after it — so the simpler grid stays the default).
static int persistent_grid() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("MV_VOL_PERSIST_WG_PER_CU");
        const int per_cu = e ? atoi(e) : 0;
        int dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
        v = (per_cu * cus) / 8 * 8;
    }
    return v;
}

// MV_VOL_EXTRA_LDS=<bytes> of unused dynamic LDS per workgroup: 12288 drops the GEMM from 4 to 3 workgroups per CU, leaving
// registers / LDS / wave slots for the latency-bound kernels of the other streams (A/B knob, default 0)
static unsigned gemm_extra_lds() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MV_VOL_EXTRA_LDS"); v = e ? atoi(e) : 0; }
    return (unsigned)v;
}

// Grid of the scheduled fp32 kernel: 2 workgroups per CU (measured optimum), a multiple of 8 so that "id % 8 == XCD" holds.
// MV_VOL_SCHED=0 switches back to the one-tile-per-workgroup kernel (A/B knob); MV_VOL_SCHED_WG_PER_CU=<k> changes the 2.
static int sched_slots() {
    static int v = -1;
    if (v < 0) {
        const char* off = getenv("MV_VOL_SCHED");
        const char* e = getenv("MV_VOL_SCHED_WG_PER_CU");
        const int per_cu = (off && atoi(off) == 0) ? 0 : (e ? atoi(e) : 2);
        int dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
        v = (per_cu * cus) / 8 * 8;
    }
    return v;
}

// Scheduled fp32 kernel: schedule width (items per round) and residency.  MV_VOL_SCHED=0 -> one-tile-per-workgroup kernel;
// MV_VOL_WALK=static -> persistent static walk; MV_VOL_WG_PER_CU=<k> residency of the default form (2).
static int vol_walk_static() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MV_VOL_WALK"); v = (e && strcmp(e, "static") == 0) ? 1 : 0; }
    return v;
}
static unsigned vol_lds_bytes() {   // dynamic LDS request that admits exactly k workgroups of this kernel per CU (160 KB LDS)
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("MV_VOL_WG_PER_CU");
        const int k = e ? atoi(e) : 2;
        const int used = 2 * 16 * 256 * 4;
        v = used;
        if (k >= 1 && k < 5) { const int need = 160 * 1024 / (k + 1) + 1024; v = need > used ? need : used; }
    }
    return (unsigned)v;
}

// name of the kernel the last mv_corr_volume call of this thread dispatched (tests assert the dispatch; bench.py names the kernel
// its roofline line is about)
static thread_local const char* g_last_vol_kernel = "";
extern "C" const char* mv_corr_volume_last_kernel(void) { return g_last_vol_kernel; }
void mv_note_volume_kernel(const char* name) { g_last_vol_kernel = name; }   // (library-internal: corr_volume_split.hip)
#define MV_VOL_KERNEL(name) (g_last_vol_kernel = (name))

// the streaming kernels number their (pair, 128-row band, 64-column sub-tile) items with an int
static bool stream_items_fit(int B, int N1, int N2) {
    return (size_t)B * (size_t)((N1 + 127) / 128) * (size_t)(N2 / 64) < ((size_t)1 << 31);
}

#ifdef MV_HS_STAMPS
extern "C" int mv_hs_probe_stamps(long long* dev_buf) {   // (probe builds only) dev_buf: 32 * 4 * 48 * 4 int64, or NULL to switch the stamps off
    return hipMemcpyToSymbol(HIP_SYMBOL(g_hs_stamps), &dev_buf, sizeof(dev_buf)) == hipSuccess ? 0 : -1;
}
#endif

// the 16-bit streaming kernel's domain and launch (shared by mv_corr_volume and mv_corr_volume_out16)
static bool h_stream_supported(int B, int C, int N1, int N2) {
    return (C == 256 || C == 128) && (N2 % 64) == 0 && ((size_t)N1 * N2 * B) >= ((size_t)1 << 22) &&
           ((size_t)N1 * N2) < ((size_t)1 << 30) &&   // (32-bit byte offsets inside a pair's block of the output)
           stream_items_fit(B, N1, N2);               // (the item index T = B * bands * sub-tiles is an int)
}

static int launch_h_stream(const uint16_t* a, const uint16_t* b, void* outp, bool out16, bool bf, int B, int C, int N1, int N2, hipStream_t s) {
    float* out = reinterpret_cast<float*>(outp);
    // CU count and the kernels' LDS attribute: per device ordinal (a process-wide static mishandled a second GPU)
    static std::atomic<int> cus_dev[64];
    static std::atomic<bool> attr_dev[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int cus = cus_dev[dev].load(std::memory_order_relaxed);
    if (!cus) {
        hipDeviceProp_t prop;
        cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        cus_dev[dev].store(cus, std::memory_order_relaxed);
    }
    static int wgs = 0;    // workgroups per CU; LDS is padded so that exactly this many are resident
    if (!wgs) { const char* e = getenv("MV_H_STREAM_WGS"); wgs = e ? atoi(e) : 2; if (wgs < 1 || wgs > 4) wgs = 2; }
    const unsigned ring = (unsigned)(2 * 64 * (C / 8) * 16);            // 2-slot B ring
    const unsigned lds = std::max(ring, (unsigned)(160 * 1024 / (wgs + 1) + 1024));
    if (!attr_dev[dev].load(std::memory_order_acquire)) {
#define MV_HS_ATTR(...) (void)hipFuncSetAttribute((const void*)corr_volume_h_stream<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
        MV_HS_ATTR(true, 16); MV_HS_ATTR(false, 16); MV_HS_ATTR(true, 8); MV_HS_ATTR(false, 8);
        MV_HS_ATTR(true, 16, true); MV_HS_ATTR(false, 16, true); MV_HS_ATTR(true, 8, true); MV_HS_ATTR(false, 8, true);
#undef MV_HS_ATTR
        attr_dev[dev].store(true, std::memory_order_release);
    }
    const dim3 g((cus & ~7) * wgs), blk(256);                           // a multiple of 8: one run per XCD
    // column regions: one region's B rows (nc / R sub-tiles x 64 rows x 2C bytes) <= 2 MB of an XCD's 4 MB L2 (measured: 640x480
    // 1 region (2.4 MB) 47.5 us, 2 regions 41.2, 3: 43.6; 1280x720 4 regions (1.8 MB) 324 us, 6 regions 345)
    static int regs_env = -1;  // MV_H_STREAM_REGIONS: A/B knob
    if (regs_env < 0) { const char* e = getenv("MV_H_STREAM_REGIONS"); regs_env = e ? atoi(e) : 0; }
    const int nc = N2 / 64;
    int R = regs_env > 0 ? regs_env : (int)(((size_t)nc * 64 * C * 2 + (2u << 20) - 1) / (2u << 20));
    R = std::max(1, std::min(R, nc));
    MV_VOL_KERNEL(out16 ? "corr_volume_h_stream<out16>" : "corr_volume_h_stream");
#define MV_HS_GO(...) hipLaunchKernelGGL((corr_volume_h_stream<__VA_ARGS__>), g, blk, lds, s, a, b, out, N1, N2, B, R)
    if (out16) {
        if (C == 256) { if (bf) MV_HS_GO(true, 16, true); else MV_HS_GO(false, 16, true); }
        else { if (bf) MV_HS_GO(true, 8, true); else MV_HS_GO(false, 8, true); }
    } else {
        if (C == 256) { if (bf) MV_HS_GO(true, 16); else MV_HS_GO(false, 16); }
        else { if (bf) MV_HS_GO(true, 8); else MV_HS_GO(false, 8); }
    }
#undef MV_HS_GO
    return mv_launch_status();
}

extern "C" int mv_corr_volume(const void* f1, const void* f2, float* out, int B, int C, int N1, int N2,
                              int in_dtype, int layout, mvStream_t stream) {
    MV_CHECK_ARG(f1 && f2 && out);
    MV_CHECK_ARG(B > 0 && C > 0 && N1 > 0 && N2 > 0);
    MV_CHECK_ARG(layout == MV_LAYOUT_CHW || layout == MV_LAYOUT_HWC);
    MV_CHECK_ARG(((uintptr_t)f1 & 15) == 0 && ((uintptr_t)f2 & 15) == 0);
    if (B > 65535) return MV_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int tiles_m = mv_ceil_div(N1, BM), tiles_n = mv_ceil_div(N2, BN);
    dim3 grid(tiles_m * tiles_n, 1, B), block(256);
    if (in_dtype == MV_F32) {
        if (C % 16) return MV_ERR_UNSUPPORTED;
        const float* a = (const float*)f1;
        const float* b = (const float*)f2;
        if (layout == MV_LAYOUT_CHW) {
            VolSched vs;
            const int slots = sched_slots();
            if (slots > 0 && N1 == N2 && (C % 32) == 0 && make_vol_sched(N1, B, slots, vs)) {
                if (vol_walk_static()) {
                    MV_VOL_KERNEL("corr_volume_f32_sched");
                    hipLaunchKernelGGL(corr_volume_f32_sched, dim3(slots), block, 0, s, a, b, out, C, N1, vs);
                } else {
                    // LDS-DMA staging (default): 2 = BK 32 x 2 stages, 3 / 4 = BK 16 x 3 / 4 stages; MV_VOL_DMA=16 -> 3, =0 -> the
                    // register-staged tile.  Measured (profiles/probes/fvol_probe.py, 640x480, B = 2, alone): register-staged 185.8 us,
                    // BK 16 x 3 stages 182.5 (x 4 stages: 182.5), BK 32 x 2 stages 175.7 us = 0.854 of the fp32 MFMA peak (1280x720: 0.878);
                    // in the frame pipeline 3.43-3.45 k / 3.47-3.54 k / 3.39 k frames/s for BK 32 / BK 16 / register-staged at one lane
                    // (the 128 KB of LDS two BK-32 workgroups hold leave the co-running small kernels 32 KB per CU) and
                    // 4.52 k / 4.48 k / 4.32 k at three lanes.
                    static int dma = -1;
                    if (dma < 0) {
                        const char* e = getenv("MV_VOL_DMA");
                        const int v = e ? atoi(e) : 32;
                        dma = v == 0 ? 0 : v == 16 ? 3 : v == 4 ? 4 : 2;
                    }
                    if (dma && (dma == 2 ? (C % 32 == 0 && C >= 64) : C >= 16 * (dma - 1))) {
                        const unsigned ldsd = std::max(vol_lds_bytes(), (unsigned)(dma == 2 ? 4 : dma) * 16 * 256 * 4);
                        static bool attr_dma = false;
                        if (!attr_dma) {
                            (void)hipFuncSetAttribute((const void*)corr_volume_f32_mixed_dma<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                            (void)hipFuncSetAttribute((const void*)corr_volume_f32_mixed_dma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                            (void)hipFuncSetAttribute((const void*)corr_volume_f32_mixed_dma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                            attr_dma = true;
                        }
                        MV_VOL_KERNEL("corr_volume_f32_mixed_dma");
                        if (dma == 2)
                            hipLaunchKernelGGL(corr_volume_f32_mixed_dma<2>, dim3((vs.R_b + vs.R_m + vs.R_s) * slots), block, ldsd, s, a,
                                               b, out, C, N1, vs, slots);
                        else if (dma == 4)
                            hipLaunchKernelGGL(corr_volume_f32_mixed_dma<4>, dim3((vs.R_b + vs.R_m + vs.R_s) * slots), block, ldsd, s, a,
                                               b, out, C, N1, vs, slots);
                        else
                            hipLaunchKernelGGL(corr_volume_f32_mixed_dma<3>, dim3((vs.R_b + vs.R_m + vs.R_s) * slots), block, ldsd, s, a,
                                               b, out, C, N1, vs, slots);
                        return mv_launch_status();
                    }
                    const unsigned lds = vol_lds_bytes();
                    static bool attr_set = false;
                    if (!attr_set && lds > 48 * 1024) {
                        (void)hipFuncSetAttribute((const void*)corr_volume_f32_mixed, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                        attr_set = true;
                    }
                    MV_VOL_KERNEL("corr_volume_f32_mixed");
                    hipLaunchKernelGGL(corr_volume_f32_mixed, dim3((vs.R_b + vs.R_m + vs.R_s) * slots), block, lds, s, a, b, out, C,
                                       N1, vs, slots);
                }
            } else if ((N1 % 4 == 0) && (N2 % 4 == 0)) {
                MV_VOL_KERNEL("corr_volume_f32_chw");
                const int pg = persistent_grid();
                if (pg > 0 && tiles_m * tiles_n * B > pg)
                    hipLaunchKernelGGL((corr_volume_f32_chw<true, true>), dim3(pg), block, 0, s, a, b, out, C, N1, N2,
                                       tiles_m, tiles_n, B);
                else
                    hipLaunchKernelGGL((corr_volume_f32_chw<true, false>), grid, block, gemm_extra_lds(), s, a, b, out, C, N1, N2,
                                       tiles_m, tiles_n, B);
            } else {
                MV_VOL_KERNEL("corr_volume_f32_chw");
                hipLaunchKernelGGL((corr_volume_f32_chw<false, false>), grid, block, 0, s, a, b, out, C, N1, N2, tiles_m,
                                   tiles_n, B);
            }
        } else {
            MV_VOL_KERNEL("corr_volume_f32_hwc");
            hipLaunchKernelGGL(corr_volume_f32_hwc, grid, block, 0, s, a, b, out, C, N1, N2, tiles_m, tiles_n);
        }
    } else if (in_dtype == MV_BF16X3) {
        // f1 / f2 = planes produced by mv_split_bf16x3: [3][B][N][C] bf16
        if (layout != MV_LAYOUT_HWC || (C % 32)) return MV_ERR_UNSUPPORTED;
        MV_VOL_KERNEL("corr_volume_bf16x3_hwc<3>");
        hipLaunchKernelGGL(corr_volume_bf16x3_hwc<3>, grid, block, 0, s, (const uint16_t*)f1, (const uint16_t*)f2, out, C,
                           N1, N2, B, tiles_m, tiles_n);
    } else if (in_dtype == MV_BF16X2) {
        // same planes, only the two leading pieces are read
        if (layout != MV_LAYOUT_HWC || (C % 32)) return MV_ERR_UNSUPPORTED;
        MV_VOL_KERNEL("corr_volume_bf16x3_hwc<2>");
        hipLaunchKernelGGL(corr_volume_bf16x3_hwc<2>, grid, block, 0, s, (const uint16_t*)f1, (const uint16_t*)f2, out, C,
                           N1, N2, B, tiles_m, tiles_n);
    } else if (in_dtype == MV_F16 || in_dtype == MV_BF16) {
        const uint16_t* a = (const uint16_t*)f1;
        const uint16_t* b = (const uint16_t*)f2;
        const bool bf = in_dtype == MV_BF16;
        if (layout == MV_LAYOUT_HWC) {
            if (C % 32) return MV_ERR_UNSUPPORTED;
            static int hstream = -1;   // MV_H_STREAM=0: the tile form below (A/B knob)
            if (hstream < 0) { const char* e = getenv("MV_H_STREAM"); hstream = (e && atoi(e) == 0) ? 0 : 1; }
            if (hstream && h_stream_supported(B, C, N1, N2)) return launch_h_stream(a, b, out, false, bf, B, C, N1, N2, s);
            MV_VOL_KERNEL("corr_volume_h_hwc");
            static int hbk = -1;
            if (hbk < 0) { const char* e = getenv("MV_H_BK"); hbk = e ? atoi(e) : 32; }
            if (hbk == 64 && (C % 64) == 0) {
                if (bf) hipLaunchKernelGGL((corr_volume_h_hwc<true, 64>), grid, block, 0, s, a, b, out, C, N1, N2, tiles_m, tiles_n);
                else hipLaunchKernelGGL((corr_volume_h_hwc<false, 64>), grid, block, 0, s, a, b, out, C, N1, N2, tiles_m, tiles_n);
            } else {
                if (bf) hipLaunchKernelGGL((corr_volume_h_hwc<true, 32>), grid, block, 0, s, a, b, out, C, N1, N2, tiles_m, tiles_n);
                else hipLaunchKernelGGL((corr_volume_h_hwc<false, 32>), grid, block, 0, s, a, b, out, C, N1, N2, tiles_m, tiles_n);
            }
        } else {
            if (C % 32) return MV_ERR_UNSUPPORTED;
            MV_VOL_KERNEL("corr_volume_h_chw");
            if (bf)
                hipLaunchKernelGGL(corr_volume_h_chw<true>, grid, block, 0, s, a, b, out, C, N1, N2, tiles_m, tiles_n);
            else
                hipLaunchKernelGGL(corr_volume_h_chw<false>, grid, block, 0, s, a, b, out, C, N1, N2, tiles_m, tiles_n);
        }
    } else {
        return MV_ERR_INVALID_ARG;
    }
    return mv_launch_status();
}

// A5, Fast mode: the volume in the ENCODER's 16-bit type (what `einsum` returns for fp16 / bf16 feature maps; flownet.py:27 widens it afterwards).
// One rounding in the GEMM's epilogue, 2-byte cells: [B, N1, N2] of in_dtype.  HWC feature maps, C = 128 / 256, N2 % 64 == 0 (the streaming
// kernel's domain): MV_ERR_UNSUPPORTED otherwise — callers then use mv_corr_volume + a cast.  Read it with mv_corr_lookup_vol16 (fp16).
extern "C" int mv_corr_volume_out16_supported(int B, int C, int N1, int N2, int in_dtype, int layout) {
    return (in_dtype == MV_F16 || in_dtype == MV_BF16) && layout == MV_LAYOUT_HWC && B > 0 && N1 > 0 && N2 > 0 && h_stream_supported(B, C, N1, N2);
}
extern "C" int mv_corr_volume_out16(const void* f1, const void* f2, void* out, int B, int C, int N1, int N2, int in_dtype, int layout,
                                    mvStream_t stream) {
    MV_CHECK_ARG(f1 && f2 && out && B > 0 && C > 0 && N1 > 0 && N2 > 0);
    MV_CHECK_ARG(((uintptr_t)f1 & 15) == 0 && ((uintptr_t)f2 & 15) == 0 && ((uintptr_t)out & 3) == 0);
    if (!mv_corr_volume_out16_supported(B, C, N1, N2, in_dtype, layout)) return MV_ERR_UNSUPPORTED;
    return launch_h_stream((const uint16_t*)f1, (const uint16_t*)f2, out, true, in_dtype == MV_BF16, B, C, N1, N2, (hipStream_t)stream);
}

extern "C" int mv_split_bf16x3(const float* x, void* planes, size_t n, mvStream_t stream) {
    MV_CHECK_ARG(x && planes && n > 0 && (n % 4) == 0);
    MV_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)planes & 7) == 0);
    const size_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(split3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)planes, n4);
    return mv_launch_status();
}
