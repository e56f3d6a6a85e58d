// A5 — all-pairs cost volume of fp32 feature maps on the 16-bit matrix pipe:  SPLIT + STREAMING form  (round 3)
//
//   out[b,i,j] = sum_c f1[b,c,i] * f2[b,c,j]      (FlowFormer MemoryEncoder.corr, reached from
//                                                   Module/Network/FlowFormerCov/flownet.py:26; `.float()` at :27)
//
// Why.  The exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the 16-bit MFMA rate: 23.6 GFLOP per 640x480 frame = 150 us
// at its 157 TFLOP/s peak, and corr_volume_f32_mixed_dma sits at 0.85 of that.  gfx950 has no TF32 path, but the reference itself
// runs this GEMM in TF32 (torch.backends.cuda.matmul.allow_tf32 / set_float32_matmul_precision("medium"),
// Module/Frontend/Frontend.py:275-277) or fp16 (MACVO_Fast.yaml:69-76), i.e. with 11-bit operands.  Here every fp32 operand is
// split into 16-bit pieces whose sum is the operand to ~2^-24 (three bf16 pieces, 8 + 8 + 8 bits: x = p0 + p1 + p2, residuals
// exact) and the fp32 product is rebuilt from the six piece products with i + j <= 2, accumulated in fp32, smallest first:
// the dropped terms are O(2^-24 |a||b|), the class of fp32's own rounding; the parity bar is the exact path's
// (|out - einsum_f64| <= 2e-5 sqrt(C), tests/test_gpu_corr.py).  Six products at the 2.5 PFLOP/s 16-bit rate = 57 us of matrix work
// per frame next to 29-40 us of output stores (184 MB): the kernel is shaped like corr_volume_h_stream — a fill kernel with a GEMM
// attached — with three operand planes.
//
// Operand PACK (mv_volume_pack, one launch for both feature maps).  fp32 [B,C,N] (CHW, as the reference hands it over) or [B,N,C]
// is split and written in exactly the order the GEMM consumes it:
//     packed[b][rb][ks][piece][lane][8 x 16-bit]      rb = row / 32, ks = k / 16, lane = (k / 8 % 2) * 32 + row % 32
// i.e. one 1-KB unit per (32 rows, 16 k, piece) = one v_mfma_f32_32x32x16 operand fragment for all 64 lanes.  A fragments are
// then ONE coalesced 1-KB global_load_dwordx4 each, B sub-tiles are contiguous runs that the LDS-DMA copies verbatim (no swizzle:
// lane-linear 16-B reads are conflict-free), and the GEMM needs neither a layout requirement nor any VALU work.  Rows past N
// repeat row N - 1 and every pair carries one extra row block holding 32 copies of row N - 1: waves beyond the bottom edge compute
// duplicates of the last row and store them ON TOP of it (identical values) instead of branching around their stores, which keeps
// the hand-counted vmcnt arithmetic identical for all waves.  The pack moves 20 MB in and 30 MB out (L2 / Infinity-Cache traffic)
// and is issued by the frame driver on another stream one frame ahead, off the GEMM stream's critical path.
//
// GEMM (corr_volume_split_stream).  Persistent 256-thread workgroups, ONE per CU and one wave per SIMD (the whole-K fragments of
// three planes are 192 registers: AGPRs), item = 128-row band x 64-column sub-tile:
//   * a wave owns 32 rows: its A fragments (KS k-steps x NP pieces x 4 registers) stay in registers for a whole band segment;
//   * B sub-tiles stream through a 3-slot LDS ring in K HALVES (64 columns x C/2 k x NP pieces = 48 KB) by LDS-DMA, two halves
//     ahead of the MFMAs; one barrier per half;
//   * per k-step 2 x NQ MFMAs (NQ = 6 piece products, 2 column blocks) against 2 x NP ds_read_b128 and 2 output stores: ~1 filler
//     per MFMA, so a single wave per SIMD keeps the matrix pipe issuing back to back;
//   * the 32 stores of item j are interleaved with the MFMAs of item j + 1 (two accumulator sets), DMA / stores / A loads are
//     inline asm with hand-counted s_waitcnt vmcnt (in-order counter) exactly as in corr_volume_h_stream;
//   * XCD-aware item order (pair, column region, band, sub-tile): one region's B planes (<= 2 MB) stay in the XCD's L2.
#include "common.h"
#include "vol_asm.h"
#include <type_traits>
#include <algorithm>
#include <atomic>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

// ------------------------------------------------------------------------------------------------ operand pack
// One 256-thread workgroup = one (operand, pair b, row block rb): wave w handles k-steps w, w + 4, ..; one thread = one 16-byte chunk
// per piece: lane = kh * 32 + li holds k = 16 ks + 8 kh + e of row 32 rb + li.  A wave writes NP contiguous 1-KB units per k-step;
// for CHW input its loads are 8 x (2 x 128 B) coalesced rows, for HWC 2 x float4 per lane.
// F16 (two fp16 pieces, 11 + 11 bits): fp16 has 5 exponent bits, so every ROW (one pixel's feature vector) is first scaled by a
// power of two 2^sh that puts its largest magnitude into [2^14, 2^15) — exact, and undone exactly by the GEMM's epilogue
// (v_ldexp_f32 by -(sh_i + sh_j)); the row's `sh` goes into the int32 table behind the units.  With the scale the pieces carry
// x to 2^-22 |x| + 2^-40 max_k|x_k| whatever the magnitude of the row: `sh` is NOT clamped (round 4) — any finite non-zero row of the fp32
// range, denormal maxima included (sh up to 14 + 149), lands in [2^14, 2^15), so no row can overflow fp16; v_ldexp_f32 takes the full int.
template <int NP, bool F16>
__global__ __launch_bounds__(256) void volume_pack_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                          uint16_t* __restrict__ p1, uint16_t* __restrict__ p2, int B, int C,
                                                          int N1, int N2, int hwc, int tile_w) {
    __shared__ float wmax[4][32];
    const int op = blockIdx.y;
    const float* __restrict__ f = op ? f2 : f1;
    uint16_t* __restrict__ out = op ? p2 : p1;
    const int N = op ? N2 : N1;
    const int KS = C >> 4, nrb = ((N + 31) >> 5) + 1;           // + the replica block of row N - 1
    if ((int)blockIdx.x >= B * nrb) return;
    const int b = blockIdx.x / nrb, rb = blockIdx.x - b * nrb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, kh = lane >> 5;
    const int row = rb == nrb - 1 ? N - 1 : min(rb * 32 + li, N - 1);
    // tile_w > 0 (mv_volume_pack_tiled): operand 2's pixels are packed in 4 x 4-cell tile order of its tile_w-pixel-wide map — packed
    // row r' = (tile, cell): the GEMM's output columns then ARE that order, i.e. every query's slice of the volume comes out tiled
    // (16 consecutive floats = one 4 x 4 block of target pixels = one 64-byte line) with the GEMM and its coalesced stores untouched.
    // The workgroup still READS 32 consecutive pixels (coalesced for CHW input) and scatters on the write side: pixel (y, x) goes to
    // packed row r' = ((y / 4) * (W / 4) + x / 4) * 16 + (y % 4) * 4 + x % 4 — four consecutive pixels stay four consecutive rows
    // (64-byte runs).  (Permuting on the read side instead cost the B = 64 pack 2x and the 32-lane step 4 %.)
    int rb_d = rb, li_d = li;
    if (tile_w > 0 && op == 1 && rb != nrb - 1 && rb * 32 + li < N) {
        const int r = rb * 32 + li, y = r / tile_w, xx = r - y * tile_w;
        const int rd = (((y >> 2) * (tile_w >> 2) + (xx >> 2)) << 4) + ((y & 3) << 2) + (xx & 3);
        rb_d = rd >> 5;
        li_d = rd & 31;
    }
    constexpr int MAXKS = 8;                                     // k-steps per wave held in registers (C <= 512)
    float x[MAXKS][8];
    float m = 0.f;
#pragma unroll
    for (int q = 0; q < MAXKS; ++q) {
        const int ks = wave + 4 * q;
        if (ks < KS) {
            const int k0 = ks * 16 + kh * 8;
            if (hwc) {
                const f32x4* src = reinterpret_cast<const f32x4*>(f + ((size_t)b * N + row) * C + k0);
                const f32x4 lo = src[0], hi = src[1];
#pragma unroll
                for (int e = 0; e < 4; ++e) { x[q][e] = lo[e]; x[q][4 + e] = hi[e]; }
            } else {
                const float* src = f + ((size_t)b * C + k0) * N + row;
#pragma unroll
                for (int e = 0; e < 8; ++e) x[q][e] = src[(size_t)e * N];
            }
            if (F16) {
#pragma unroll
                for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(x[q][e]));
            }
        }
    }
    int sh = 0;
    if (F16) {
        m = fmaxf(m, __shfl_xor(m, 32, 64));                     // the two k halves of the row
        if (kh == 0) wmax[wave][li] = m;
        __syncthreads();
        m = fmaxf(fmaxf(wmax[0][li], wmax[1][li]), fmaxf(wmax[2][li], wmax[3][li]));
        if (m > 0.f && m < INFINITY) sh = 14 - ilogbf(m);                        // NaN / inf / all-zero rows: unscaled
        int* exps = reinterpret_cast<int*>(reinterpret_cast<char*>(out) + (size_t)B * nrb * KS * NP * 1024);
        if (wave == 0 && kh == 0) exps[((size_t)b * nrb + rb_d) * 32 + li_d] = sh;
    }
#pragma unroll
    for (int q = 0; q < MAXKS; ++q) {
        const int ks = wave + 4 * q;
        if (ks < KS) {
            s16x8 pc[NP];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float r = F16 ? ldexpf(x[q][e], sh) : x[q][e];
#pragma unroll
                for (int p = 0; p < NP; ++p) {                  // piece p = round-to-nearest of what is left; residual exact in fp32
                    if (F16) {
                        const _Float16 h = (_Float16)r;
                        pc[p][e] = __builtin_bit_cast(short, h);
                        r -= (float)h;
                    } else {
                        const __bf16 h = (__bf16)r;
                        pc[p][e] = __builtin_bit_cast(short, h);
                        r -= (float)h;
                    }
                }
            }
            s16x8* dst = reinterpret_cast<s16x8*>(out) + (((size_t)b * nrb + rb_d) * KS + ks) * NP * 64 + kh * 32 + li_d;
#pragma unroll
            for (int p = 0; p < NP; ++p) dst[p * 64] = pc[p];
        }
    }
}

// ------------------------------------------------------------------------------------------------ GEMM
// (asm loads / stores below work on accumulator-file registers: on gfx90a+ VMEM data operands and MFMA A/B sources may be AGPRs)
template <int NP, bool F16, int KS, int NWV>
struct SplitCfg {
    static_assert(NWV == 4 || NWV == 8, "waves per workgroup");
    static constexpr int JB = 8 / NWV;                 // 32-column blocks of the sub-tile a wave owns (NWV = 8: one, 2 waves per SIMD)
    static constexpr int KH = KS / 2;                  // k-steps per K half
    static constexpr int NA = KS * NP;                 // A fragment units per wave and band
    // ... of which this many live in the accumulator register file.  Three planes (192 registers) + 2 x 32 accumulators would fill
    // all 256 AGPRs: hipcc then has no register for the copy it makes of an accumulator element on its way into an asm store and
    // evicts part of an A unit for it — also during the flush, while that unit's load is still in flight (seen in the ISA: v_accvgpr_read
    // of a72..a75 behind the loads, v_accvgpr_write a72 in front of every store).  The last two units therefore live in VGPRs.
    static constexpr int NA_ACC = NWV == 8 ? NA - 8 : (NP == 3 ? NA - 2 : NA);
    static constexpr int HALF_UNITS = KH * NP;         // 1-KB units of one row block in one K half
    static constexpr int SLOT_BYTES = 2 * HALF_UNITS * 1024;   // 64 columns = 2 row blocks
    static constexpr int NSLOT = 3;
    static constexpr int D = 2 * HALF_UNITS / NWV;     // DMA pieces per wave and half
    static constexpr int NQ = NP == 3 ? 6 : 3;         // piece products
    static constexpr int NM = NQ * JB;                 // MFMAs (= filler slots) per k-step
    static constexpr int SPH = KH * JB;                // output stores per wave and half: JB behind each k-step
    static constexpr int FLUSH = 2 * SPH;              // stores of one item
    // LDS-DMA pieces of a half: PH straight behind the barrier, the rest spread over the k-steps (at most 2 per k-step)
    static constexpr int PH = D >= 12 ? 3 : 2;
    static constexpr int pieces_in(int ks) { return (D - PH) / KH + (ks < (D - PH) % KH ? 1 : 0); }
    static constexpr int first_piece(int ks) { return PH + ks * ((D - PH) / KH) + (ks < (D - PH) % KH ? ks : (D - PH) % KH); }
    static constexpr int last_piece_ks() { int k = 0; for (int i = 0; i < KH; ++i) if (pieces_in(i) > 0) k = i; return k; }
    // filler slots of a k-step: reads [0, JB NP), stores [JB NP, JB NP + JB), then up to two pieces (sharing the last slot with a
    // store when the k-step has no slot left: NP = 2)
    static constexpr int SL_STORE = JB * NP;
    static constexpr int SL_PIECE0 = SL_STORE + JB < NM ? SL_STORE + JB : NM - 1;
    static constexpr int SL_PIECE1 = SL_PIECE0 + 2 < NM ? SL_PIECE0 + 2 : -1;
    static_assert(SL_STORE + JB <= NM, "a k-step must have a slot for each read and store");
    static_assert(SL_PIECE1 >= 0 || pieces_in(0) <= 1, "two pieces per k-step need two slots");
    // vmcnt arithmetic (in-order counter).  The pieces of half h are issued inside half h - 2; its last piece goes out in k-step
    // last_piece_ks(), BEHIND that k-step's stores.  Behind that last piece the wave issues: the stores of the remaining
    // k-steps of half h - 2, then the D pieces and SPH stores of half h - 1.  "vmcnt <= that number" at the barrier of half h
    // therefore means the pieces have landed.  Without stores in flight (first / second item of a segment) the counts shrink.
    static constexpr int STORES_BEHIND_LAST_PIECE = JB * (KH - 1 - last_piece_ks());
    static constexpr int EB = F16 ? JB : 0;            // F16: column-exponent loads at the head of every item's first half
    static constexpr int W_STEADY0 = STORES_BEHIND_LAST_PIECE + D + SPH;        // first half of an item: the half before was a second half
    static constexpr int W_STEADY = STORES_BEHIND_LAST_PIECE + D + SPH + EB;    // second half: the half before carried the EB loads
    static_assert(W_STEADY <= 63 && 2 * (D + SPH) + EB <= 63 && FLUSH + D + EB <= 63, "vmcnt is a 6-bit counter");
    static_assert(KS % 2 == 0 && (2 * HALF_UNITS) % NWV == 0 && HALF_UNITS % D == 0, "a wave's pieces lie inside one row block");
    // EARLY START of a band segment (FIRSTK = 1: band change, the previous segment's FLUSH stores ride behind the A loads;
    // FIRSTK = 2: kernel start, the A loads ride behind the first two halves of B pieces).  The first item does not wait for all NA
    // fragment units: in front of k-step ks of its first half it waits for the units of k-steps 0 .. ks only — "at most as many
    // operations outstanding as were issued BEHIND that unit" (in-order counter; never more than VM_MAX can be in flight, so a
    // larger count needs no wait at all and is clamped) — and the rest lands while the first k-steps are multiplied.
    static constexpr int VM_MAX = 60;                  // (the counter has 6 bits; a margin of 3)
    static constexpr int EARN = F16 ? 4 : 0;           // row-exponent loads behind the A units
    static constexpr int clampw(int x) { return x < VM_MAX ? x : VM_MAX; }
    static constexpr int behind_unit(int firstk, int ks) { return NA - NP * (ks + 1) + EARN + (firstk == 1 ? FLUSH : 0); }
    static constexpr int first_wait(int firstk, int ks) { return clampw(behind_unit(firstk, ks)); }
    // barrier waits of the first item: half 0 needs its B pieces (issued before the A units: NA + EARN (+ FLUSH | + D: the other
    // half's pieces) operations behind them), half 1 additionally every A unit and row exponent
    static constexpr int WB0_CHANGE = clampw(NA + EARN + FLUSH), WB1_CHANGE = FLUSH + D + EB;
    static constexpr int WB0_START = clampw(D + NA + EARN), WB1_START = D + EB;
};

#ifndef MV_SPLIT_DMA_IMM
#define MV_SPLIT_DMA_IMM 1          // LDS-DMA pieces in groups of four behind one M0 / base (round 4; A/B: profiles/r04_split_variants_ab1.log)
#endif
#ifndef MV_SPLIT_VACC
#define MV_SPLIT_VACC 1             // f16x2: accumulators in VGPRs (asm MFMAs), true ping-pong between the two sets, no v_accvgpr_read
#endif
#ifdef MV_SPLIT_PROBE
#define MV_SPLIT_PROBE_SLACK 2      // the stamp store of the probe build is one more memory operation per item
#else
#define MV_SPLIT_PROBE_SLACK 0
#endif
// Timing probes (profiles/probes/split_variants.sh builds the library with -DMV_SPLIT_PROBE): MV_SPLIT_DBG=<bits> knocks pieces of the
// kernel out at run time — 1 stores, 2 LDS-DMA, 4 barriers + waits, 8 B-fragment reads — results are then wrong by design.
#ifdef MV_SPLIT_PROBE
__constant__ int g_split_dbg = 0;
__constant__ long long* g_split_stamps = nullptr;   // [workgroup][wave][64 items][8] s_memtime stamps (MV_SPLIT_DBG bit 16)
#ifdef MV_SPLIT_KNOCK        // compile-time knock-outs (no run-time branches in the stream): -DMV_SPLIT_KNOCK=<bits>
#define DBG(bit) ((bit) == 16 ? (g_split_dbg & 16) : ((MV_SPLIT_KNOCK) & (bit)))
#else
#define DBG(bit) (g_split_dbg & (bit))
#endif
#define STAMP(i) do { if (DBG(16)) stamps[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DBG(bit) false
#define STAMP(i) ((void)0)
#endif

template <int NP, bool F16, int KS, int NWV>
__global__ __launch_bounds__(64 * NWV) __attribute__((amdgpu_waves_per_eu(NWV / 4, NWV / 4))) void corr_volume_split_stream(
    const uint16_t* __restrict__ pk1, const uint16_t* __restrict__ pk2, float* __restrict__ out, int N1, int N2, int B, int R) {
    using Cf = SplitCfg<NP, F16, KS, NWV>;
    constexpr int KH = Cf::KH, NA = Cf::NA, HU = Cf::HALF_UNITS, D = Cf::D, NSLOT = Cf::NSLOT, JB = Cf::JB;
    constexpr unsigned SLOT_BYTES = Cf::SLOT_BYTES;
    extern __shared__ __attribute__((aligned(16))) i32x4 smem_sp[];   // B ring: NSLOT slots of [2 row blocks][KH][NP][64 lanes] x 16 B
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // NWV = 4: wave w owns rows 32 w .. and both 32-column blocks of the sub-tile; NWV = 8 (two waves per SIMD: the fillers of one
    // hide behind the MFMAs of the other): waves 2 w, 2 w + 1 share rows 32 w .. and take one column block each
    const int wr = NWV == 8 ? wave >> 1 : wave;
    const int jb0 = NWV == 8 ? wave & 1 : 0;
    const int kh = lane >> 5, li = lane & 31;
    const int nb = (N1 + 127) >> 7, nc = N2 >> 6;
    const int nrb1 = ((N1 + 31) >> 5) + 1, nrb2 = ((N2 + 31) >> 5) + 1;   // row blocks per pair incl. the replica block
    const int per = nb * nc, T = B * per;                                  // T < 2^31 (host-checked)
    int it, it_end;
    {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3, nj = gridDim.x >> 3;
        const long lo = (long)x * T / 8, hi = ((long)x + 1) * T / 8;
        it = (int)(lo + (hi - lo) * j / nj);
        it_end = (int)(lo + (hi - lo) * (j + 1) / nj);
    }
    if (it >= it_end) return;
    auto reg_c0 = [&](int g) { return (int)((long)g * nc / R); };        // first sub-tile of column region g
    auto decode = [&](int i, int& b, int& g, int& band, int& c) {        // rare: once per run and per segment
        b = i / per;
        int rem = i - b * per;
        g = 0;
        while (g + 1 < R && rem >= nb * reg_c0(g + 1)) ++g;
        rem -= nb * reg_c0(g);
        const int w = reg_c0(g + 1) - reg_c0(g);
        band = rem / w;
        c = reg_c0(g) + (rem - band * w);
        b = __builtin_amdgcn_readfirstlane(b);
        g = __builtin_amdgcn_readfirstlane(g);
        band = __builtin_amdgcn_readfirstlane(band);
        c = __builtin_amdgcn_readfirstlane(c);
    };
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem_sp);   // low 32 bits of a flat LDS pointer = LDS byte offset
    const unsigned lane16 = (unsigned)lane * 16u;
    // F16: per-row scale exponents behind the units of each packed operand (volume_pack_kernel)
    const int* ex1 = reinterpret_cast<const int*>(reinterpret_cast<const char*>(pk1) + (size_t)B * nrb1 * (NA * 1024));
    const int* ex2 = reinterpret_cast<const int*>(reinterpret_cast<const char*>(pk2) + (size_t)B * nrb2 * (NA * 1024));

    // ---- B loader (LDS-DMA), two K halves ahead of the MFMAs; all walking state wave-uniform.  A half of a sub-tile is two
    // contiguous runs of HU units (row blocks 2c, 2c + 1); wave w copies units [w D, (w + 1) D) of that 2 HU-unit image to the same
    // position of the slot.
    // Walking state (round 4): the loader is EXACTLY two halves = one item ahead of the MFMAs, so the half it fetches inside consumer half
    // H is half H of the next item — a compile-time constant — and its pointer only moves once per item: + ITEM_STRIDE inside a band
    // segment (consecutive sub-tiles), back to the region's first sub-tile at a band change, `decode` (64-bit divisions) only at a region /
    // pair change.  Round 3 walked (pair, region, band, sub-tile) counters behind EVERY half: ~35 SALU and 3-4 taken branches in the
    // middle of each half's MFMA stream (seen in the ISA), i.e. a ~200-cycle hole per 1536 cycles of matrix work.
    constexpr size_t ITEM_STRIDE = (size_t)2 * (2 * HU) * 1024;        // one sub-tile = two row blocks of the packed operand
    int ld_it = it, ld_slot = 0, ld_band, ld_w, ld_seg_end;
    auto src_of = [&](int b, int c) {   // the wave's D units of a half: units [wave D, (wave + 1) D) of the 2 HU-unit slot image
        return reinterpret_cast<const char*>(pk2) + (((size_t)b * nrb2 + 2 * c + (wave * D) / HU) * (2 * HU) + (wave * D) % HU) * 1024;
    };
    const char* ld_ptr;                 // first half of the item being fetched
    const char* ld_ptr0;                // ... of the first sub-tile of its (pair, region)
    auto loader_decode = [&]() __attribute__((always_inline)) {      // rare: run start, region / pair change
        int b_, g_, c_;
        decode(ld_it, b_, g_, ld_band, c_);
        const int c0_ = reg_c0(g_);
        ld_w = reg_c0(g_ + 1) - c0_;
        ld_ptr0 = src_of(b_, c0_);
        ld_ptr = ld_ptr0 + (size_t)(c_ - c0_) * ITEM_STRIDE;
        ld_seg_end = min(it_end, ld_it + (c0_ + ld_w - c_));
    };
    loader_decode();
    auto next_item = [&]() __attribute__((always_inline)) {          // behind the last piece of an item's SECOND half
        if (ld_it + 1 < it_end) {                // past the end of the run the last item is simply fetched again
            ++ld_it;
            ld_ptr += ITEM_STRIDE;
            if (ld_it == ld_seg_end) {           // (uniform, once per segment)
                if (++ld_band < nb) {            // next band of the region: the same sub-tiles again
                    ld_ptr = ld_ptr0;
                    ld_seg_end = min(it_end, ld_it + ld_w);
                } else {
                    loader_decode();
                }
            }
        }
    };
    auto advance_slot = [&]() __attribute__((always_inline)) { ld_slot = ld_slot == NSLOT - 1 ? 0 : ld_slot + 1; };
    auto issue_half = [&](auto HH) __attribute__((always_inline)) {       // prologue form: the D pieces as one block
        constexpr int H = decltype(HH)::value;
        const char* src = ld_ptr + (size_t)H * (HU * 1024);
        const unsigned dst = lds0 + (unsigned)ld_slot * SLOT_BYTES + (unsigned)(wave * D) * 1024u;
#pragma unroll
        for (int i = 0; i < D; ++i) glds16_s(lane16, src + i * 1024, dst + (unsigned)i * 1024u);
        advance_slot();
        if (H == 1) next_item();
    };

    // ---- A fragments: NA coalesced 1-KB units of this wave's row block, whole K, all pieces -> accumulator-file registers.
    // asm like the DMA: the wait is placed by hand so that the previous segment's last 32 stores can be issued BEHIND these loads
    // and drain while the next sub-tile is multiplied (vmcnt retires in order; a compiler-tracked load would be waited for with
    // vmcnt(0)).  The destination counts as defined at the end of the asm statement; the "+a" re-definition behind the hand-placed
    // wait is where the values become usable (same scheme as corr_volume_h_stream, whose bitwise test guards it; here
    // test_corr_volume_split_* run multi-segment workgroups).
    i32x4 afr[NA];
    i32x4 ear[4];                                // F16: scale exponents of this lane's 16 accumulator rows (rows 8 g + 4 kh + 0..3)
    auto issue_a = [&](int b, int band) __attribute__((always_inline)) {
        const int rb = min(band * 4 + wr, nrb1 - 1);             // waves past the bottom edge take the replica block (row N1 - 1 x 32)
        const char* A = reinterpret_cast<const char*>(pk1) + ((size_t)b * nrb1 + rb) * (size_t)(NA * 1024);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            // 12-bit immediates reach 4 units; the rest of the offset rides on the (uniform) base.  Register file of the destination:
            // an asm output counts as defined at the end of the statement, so hipcc may MOVE it right away — before the data has
            // landed — if the file the constraint names has no room for it until its first use.  With two waves per SIMD (256
            // registers: 128 + 128) the accumulator file holds 32 accumulator registers + 24 of the 32 A units; the last 8 units
            // are therefore loaded into VGPRs in the first place (seen otherwise: v_accvgpr_read copies of half the fragments in
            // front of the wait, garbage products).  The ISA dump must show no move of `afr` between these loads and the "+" marks.
            if (i < Cf::NA_ACC) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&a"(afr[i]) : "v"(lane16), "s"(A + (i >> 2) * 4096), "n"((i & 3) * 1024) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(afr[i]) : "v"(lane16), "s"(A + (i >> 2) * 4096), "n"((i & 3) * 1024) : "memory");
        }
        if (F16) {
            const int* E = ex1 + ((size_t)b * nrb1 + rb) * 32;
            const unsigned vo = (unsigned)kh * 16u;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(ear[gq]) : "v"(vo), "s"(E), "n"(gq * 32) : "memory");
        }
    };
    int nea[16];                                 // F16: minus the row exponents, per accumulator register
    // F16: scale exponents of this lane's two output columns of an item, loaded at the head of the item's first half
    int cur_b = 0, cur_c = 0;                    // (pair, sub-tile) of the item being multiplied
    auto issue_eb = [&](int& e0, int& e1) __attribute__((always_inline)) {
        const int* E = ex2 + ((size_t)cur_b * nrb2 + 2 * cur_c + jb0) * 32;
        const unsigned vo = (unsigned)li * 4u;
        asm volatile("global_load_dword %0, %1, %2" : "=&v"(e0) : "v"(vo), "s"(E) : "memory");
        if (JB == 2) asm volatile("global_load_dword %0, %1, %2 offset:128" : "=&v"(e1) : "v"(vo), "s"(E) : "memory");
    };

    float* O = nullptr;                          // wave-uniform: column 0 of the CURRENT item's output block (row 0 of the pair)
    unsigned roff[16];                           // per-lane byte offsets of the 16 accumulator rows (C/D layout), fixed for a band segment
#ifdef MV_SPLIT_PROBE_STORE4
    unsigned toff[4];                            // (probe) lane l -> row 8 g + l / 8, 16-byte column chunk l % 8
#endif
    // one output store: accumulator row r of column block j; data straight from the accumulator file ("a": the MFMA results never
    // visit a VGPR)
    // F16: the value is first rescaled by 2^-(row exponent + column exponent) (exact: v_ldexp_f32), `ebj` = the column's exponent
    auto store_j = [&](const f32x16& pj, int r, int j, float* Ob, int ebj) __attribute__((always_inline)) {
        if (DBG(1)) {                            // (probe builds only) keep the accumulators alive without storing them
            asm volatile("" ::"a"(pj[r]));
            return;
        }
        if (F16) {
            const float v = __builtin_ldexpf(pj[r], nea[r] - ebj);
#if defined(MV_SPLIT_PROBE_STORE4)   // timing probe (wrong results): the same BYTES in a quarter of the store instructions — one dwordx4 per four accumulator rows
            if ((r & 3) == 0) {
                f32x4 v4 = {v, v, v, v};
                const unsigned off4 = toff[r >> 2];      // rows 8 g .. 8 g + 7 x 32 columns: exactly what the four dword stores of this group cover
                if (j == 0) asm volatile("global_store_dwordx4 %0, %1, %2" MV_VOL_STORE_ASM_MOD ::"v"(off4), "v"(v4), "s"(Ob) : "memory");
                else asm volatile("global_store_dwordx4 %0, %1, %2 offset:128" MV_VOL_STORE_ASM_MOD ::"v"(off4), "v"(v4), "s"(Ob) : "memory");
            } else {
                asm volatile("" ::"v"(v));
            }
#elif defined(MV_SPLIT_PROBE_STORE1OF4)   // timing probe (wrong results): a quarter of the store instructions AND a quarter of the bytes
            if ((r & 3) == 0) {
                if (j == 0) asm volatile("global_store_dword %0, %1, %2" MV_VOL_STORE_ASM_MOD ::"v"(roff[r]), "v"(v), "s"(Ob) : "memory");
                else asm volatile("global_store_dword %0, %1, %2 offset:128" MV_VOL_STORE_ASM_MOD ::"v"(roff[r]), "v"(v), "s"(Ob) : "memory");
            } else {
                asm volatile("" ::"v"(v));
            }
#else
            if (j == 0) asm volatile("global_store_dword %0, %1, %2" MV_VOL_STORE_ASM_MOD ::"v"(roff[r]), "v"(v), "s"(Ob) : "memory");
            else asm volatile("global_store_dword %0, %1, %2 offset:128" MV_VOL_STORE_ASM_MOD ::"v"(roff[r]), "v"(v), "s"(Ob) : "memory");
#endif
        } else {
            if (j == 0) asm volatile("global_store_dword %0, %1, %2" MV_VOL_STORE_ASM_MOD ::"v"(roff[r]), "a"(pj[r]), "s"(Ob) : "memory");
            else asm volatile("global_store_dword %0, %1, %2 offset:128" MV_VOL_STORE_ASM_MOD ::"v"(roff[r]), "a"(pj[r]), "s"(Ob) : "memory");
        }
    };
    int slot = 0;                                // ring slot of the half about to be multiplied
#ifdef MV_SPLIT_PROBE
    long long stamps[6] = {0, 0, 0, 0, 0, 0};
    int n_stamped = 0;
#endif

    // piece products, smallest first: (a0,b2) (a1,b1) (a2,b0) (a0,b1) (a1,b0) (a0,b0)   [NP = 2: (a0,b1) (a1,b0) (a0,b0)]
    constexpr int PA[6] = {0, 1, NP == 3 ? 2 : 0, 0, 1, 0};
    constexpr int PB[6] = {NP == 3 ? 2 : 1, NP == 3 ? 1 : 0, 0, 1, 0, 0};

    // one K half of one item: ring upkeep, KH x 2 x NQ MFMAs into (c0, c1); with PREV, SPH stores of (p0, p1) ride between them.
    // W = how many of this wave's memory operations were issued behind the DMA pieces this half consumes (see SplitCfg).
    //
    // With ONE wave per SIMD every instruction that is not an MFMA has to fit into the 32-cycle shadow of one: measured on this
    // kernel (profiles/probes/split_variants.sh), each class of filler issued in bursts cost the matrix pipe ~5 % (B-fragment reads,
    // stores, DMA pieces: 122.6 -> 117 / 116 / 115 us when knocked out, 102 us with all of them gone; in-kernel stamps: 49 cycles
    // per MFMA instead of 32).  So the fillers are placed ONE BY ONE behind individual MFMAs ("slots", 2 NQ per k-step):
    //   slots 0 .. 2 NP - 1   one B-fragment read each for the NEXT k-step,
    //   slots 2 NP, 2 NP + 1  the two output stores of the previous item this k-step carries,
    //   slots 2 NP + 2, + 4   an LDS-DMA piece of half + 2 (the D pieces are spread over the whole half; the first ones go out
    //                         straight behind the barrier, in the shadow of the first fragment reads' LDS latency).
    // the A units of k-step `ks` count as defined behind the wait that covers them (see issue_a)
    auto mark_a = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int i = ks * NP + p;
            if (i < Cf::NA_ACC) asm volatile("" : "+a"(afr[i]));
            else asm volatile("" : "+v"(afr[i]));
        }
    };
    auto first_wait = [&](auto FK, int ks) __attribute__((always_inline)) {     // (ks is a constant after unrolling)
        constexpr int F = decltype(FK)::value;
        switch (ks) {
            case 0: wait_vmcnt<Cf::first_wait(F, 0)>(); break;
            case 1: wait_vmcnt<Cf::first_wait(F, 1)>(); break;
            case 2: wait_vmcnt<Cf::first_wait(F, 2)>(); break;
            case 3: wait_vmcnt<Cf::first_wait(F, 3)>(); break;
            case 4: wait_vmcnt<Cf::first_wait(F, 4)>(); break;
            case 5: wait_vmcnt<Cf::first_wait(F, 5)>(); break;
            case 6: wait_vmcnt<Cf::first_wait(F, 6)>(); break;
            default: wait_vmcnt<Cf::first_wait(F, 7)>(); break;
        }
    };
    static_assert(KH == 8, "first_wait covers eight k-steps per half");
    auto half = [&](auto HH, auto WW, auto PREV, auto FK, f32x16& c0, f32x16& c1, const f32x16& p0, const f32x16& p1, int (&ec)[2], int (&ep)[2]) __attribute__((always_inline)) {
        constexpr int H = decltype(HH)::value;
        constexpr int W = decltype(WW)::value;
        constexpr bool HAVE_PREV = decltype(PREV)::value;
        constexpr int FIRSTK = decltype(FK)::value;
        STAMP(H * 3 + 0);
        if (!DBG(4)) wait_vmcnt_barrier<(W > 2 ? W - (MV_SPLIT_PROBE_SLACK) : W)>();    // behind the barrier all four waves' pieces are in, and everyone has left slot - 1
        STAMP(H * 3 + 1);
        const char* src = ld_ptr + (size_t)H * (HU * 1024);   // half + 2 (= half H of the next item) -> the slot everyone has just left
        const unsigned dst = lds0 + (unsigned)ld_slot * SLOT_BYTES + (unsigned)(wave * D) * 1024u;
        auto piece = [&](int pi) __attribute__((always_inline)) {   // (pi is a constant after unrolling)
#if MV_SPLIT_DMA_IMM
            // groups of four pieces share M0 and the scalar base (immediate offsets 0 / 1 / 2 / 3 KB on both addresses): 1 + 3 SALU per
            // group instead of 8 per piece (M0 save / write / restore + a 64-bit source and a 32-bit destination address each)
            if (!DBG(2)) {
                const char* gsrc = src + (pi >> 2) * 4096;
                switch (pi & 3) {
                    case 0: glds16_m0(lane16, gsrc, dst + (unsigned)(pi >> 2) * 4096u); break;
                    case 1: glds16_next<1024>(lane16, gsrc); break;
                    case 2: glds16_next<2048>(lane16, gsrc); break;
                    default: glds16_next<3072>(lane16, gsrc); break;
                }
            }
#else
            if (!DBG(2)) glds16_s(lane16, src + pi * 1024, dst + (unsigned)pi * 1024u);
#endif
            if (pi == D - 1) {
                advance_slot();
                if (H == 1) next_item();
            }
        };
        const i32x4* q = smem_sp + (unsigned)slot * (SLOT_BYTES / 16) + lane;
        i32x4 fb[2][JB][NP];                     // [stage][column block][piece]
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int p = 0; p < NP; ++p) fb[0][j][p] = q[(((jb0 + j) * KH + 0) * NP + p) * 64];
        __builtin_amdgcn_sched_barrier(0);
        if (F16 && H == 0) {
            // the previous item's column exponents were loaded in front of the pieces this barrier has just waited for: usable now
            if (HAVE_PREV) asm volatile("" : "+v"(ep[0]), "+v"(ep[1]));
            issue_eb(ec[0], ec[1]);
        }
#pragma unroll
        for (int pi = 0; pi < Cf::PH; ++pi) piece(pi);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            const int npk = Cf::pieces_in(ks), pk0 = Cf::first_piece(ks);     // (compile-time after unrolling)
            if (FIRSTK != 0) {                   // first item of a band segment: the A units arrive while it runs
                if (H == 0) first_wait(FK, ks);  // ... half 0: unit by unit; half 1: all of them are in behind its barrier
                mark_a(H * KH + ks);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int sl = 0; sl < Cf::NM; ++sl) {
                const int qd = sl / JB, jb = sl % JB;
                constexpr int base = NP == 3 ? 0 : 3;
                const int pa = PA[base + qd], pb = PB[base + qd];
                const i32x4 a = afr[(H * KH + ks) * NP + pa];
                f32x16& c = jb ? c1 : c0;
                if (F16 && NWV == 4 && MV_SPLIT_VACC) {
                    // accumulators in VGPRs (the builtin puts them into the accumulator file as soon as the kernel's asm names an "a"
                    // register; every output then costs a v_accvgpr_read, issued as a 32-instruction burst behind the item's last MFMA).
                    // As asm the two sets (x, y) are plain VGPR tuples, the ping-pong is real and the stores read them directly.
                    // Hazards hipcc no longer sees: the VALU reads (v_ldexp_f32 of the store path) come >= 4 MFMAs behind the set's last
                    // write; the flush is behind `s_nop 15; s_nop 7` already; SrcC = vDst back to back needs no wait states.
                    const i32x4 bq = fb[cur][jb][pb];
                    if (H == 0 && ks == 0 && qd == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "a"(a), "v"(bq));
                    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "a"(a), "v"(bq));
                } else if (H == 0 && ks == 0 && qd == 0) {      // the first product of an item starts its accumulators (C = 0 operand)
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.f;
                    if (F16) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, fb[cur][jb][pb]), z, 0, 0, 0);
                    else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, fb[cur][jb][pb]), z, 0, 0, 0);
                } else {
                    if (F16) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, fb[cur][jb][pb]), c, 0, 0, 0);
                    else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, fb[cur][jb][pb]), c, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- the filler(s) of this slot
                if (sl < Cf::SL_STORE) {
                    if (ks + 1 < KH) {
                        const int j2 = sl / NP, p2 = sl % NP;
                        fb[nxt][j2][p2] = DBG(8) ? fb[cur][j2][p2] : q[(((jb0 + j2) * KH + ks + 1) * NP + p2) * 64];
                    }
                } else if (sl < Cf::SL_STORE + JB) {
                    const int js = sl - Cf::SL_STORE;
                    if (HAVE_PREV) store_j(js ? p1 : p0, H * KH + ks, js, O - 64, ep[js]);   // one accumulator row per k-step and column block
                }
                if (sl == Cf::SL_PIECE0 && npk > 0) piece(pk0);
                if (sl == Cf::SL_PIECE1 && npk > 1) piece(pk0 + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        slot = slot == NSLOT - 1 ? 0 : slot + 1;
        STAMP(H * 3 + 2);
    };
    using Yes = std::true_type;
    using No = std::false_type;
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    auto item = [&](auto W0, auto W1, auto PREV, auto FK, f32x16& c0, f32x16& c1, const f32x16& p0, const f32x16& p1, int (&ec)[2], int (&ep)[2]) __attribute__((always_inline)) {
        half(H0{}, W0, PREV, FK, c0, c1, p0, p1, ec, ep);
        half(H1{}, W1, PREV, FK, c0, c1, p0, p1, ec, ep);
        ++cur_c;
#ifdef MV_SPLIT_PROBE
        if (DBG(16) && lane == 0 && n_stamped < 64) {
            long long* o = g_split_stamps + (((size_t)blockIdx.x * 4 + wave) * 64 + n_stamped) * 8;
#pragma unroll
            for (int i = 0; i < 6; ++i) o[i] = stamps[i];
            o[6] = it;
        }
        ++n_stamped;
#endif
        ++it;
        O += 64;
    };
    auto flush = [&](const f32x16& p0, const f32x16& p1, int (&ep)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            store_j(p0, r, 0, O - 64, ep[0]);
            if (JB == 2) store_j(p1, r, 1, O - 64, ep[1]);
        }
    };
    static_assert(KS == 16, "store interleave: one accumulator row per k-step and column block");
    // waits: see the derivation in SplitCfg / DESIGN.md.  first item of a segment: everything older than the 32 flush stores has
    // landed (hand wait below), no stores ride along; second item: D pieces (+ SPH stores) behind the pieces it consumes; then steady.
    using WF0 = std::integral_constant<int, Cf::WB0_CHANGE>;      // first item behind a band change
    using WF1 = std::integral_constant<int, Cf::WB1_CHANGE>;
    using WP0 = std::integral_constant<int, Cf::WB0_START>;       // first item of the kernel
    using WP1 = std::integral_constant<int, Cf::WB1_START>;
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using WS0 = std::integral_constant<int, D>;
    using WS1 = std::integral_constant<int, D + Cf::SPH + Cf::EB>;
    using WW0 = std::integral_constant<int, Cf::W_STEADY0>;
    using WW = std::integral_constant<int, Cf::W_STEADY>;

    f32x16 x0, x1, y0, y1;
    int ex[2] = {0, 0}, ey[2] = {0, 0};           // F16: column exponents of the items accumulating in (x0, x1) / (y0, y1)
    int b, g, band, c0i;
    decode(it, b, g, band, c0i);
    issue_half(std::integral_constant<int, 0>{});   // halves 0 and 1 of the first item -> slots 0, 1
    issue_half(std::integral_constant<int, 1>{});
    issue_a(b, band);                            // behind them: the first item starts on the units of its first k-steps (EARLY START)
    // Rotated loop: a segment's FIRST item sits at the bottom of the previous pass (two instantiations — kernel start / band
    // change — without a branch that would merge the loader's scalar state through phis hipcc then keeps in VGPRs).
    int seg_end;
    auto seg_setup = [&]() __attribute__((always_inline)) {
        seg_end = min(it_end, it + (reg_c0(g + 1) - c0i));
        O = out + (size_t)b * N1 * N2 + (size_t)c0i * 64 + jb0 * 32;
        cur_b = b;
        cur_c = c0i;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            roff[r] = ((unsigned)min(band * 128 + wr * 32 + 4 * kh + (r & 3) + 8 * (r >> 2), N1 - 1) * (unsigned)N2 + li) * 4u;
#ifdef MV_SPLIT_PROBE_STORE4
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            toff[gq] = ((unsigned)min(band * 128 + wr * 32 + 8 * gq + (lane >> 3), N1 - 1) * (unsigned)N2 + (unsigned)(lane & 7) * 4u) * 4u;
#endif
    };
    auto row_exponents = [&]() __attribute__((always_inline)) {   // landed with the last A units (second barrier wait of the first item)
        if (F16) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) asm volatile("" : "+v"(ear[gq]));
#pragma unroll
            for (int r = 0; r < 16; ++r) nea[r] = -ear[r >> 2][r & 3];
        }
    };
    seg_setup();
    item(WP0{}, WP1{}, No{}, K2{}, x0, x1, x0, x1, ex, ex);
    row_exponents();
    while (true) {                               // one pass per (pair, region, band) segment of the run, entered behind its first item
        bool in_y = false;
        if (it < seg_end) {
            item(WS0{}, WS1{}, Yes{}, K0{}, y0, y1, x0, x1, ey, ex);
            in_y = true;
            while (it + 2 <= seg_end) {
                item(WW0{}, WW{}, Yes{}, K0{}, x0, x1, y0, y1, ex, ey);
                item(WW0{}, WW{}, Yes{}, K0{}, y0, y1, x0, x1, ey, ex);
            }
            if (it < seg_end) {
                item(WW0{}, WW{}, Yes{}, K0{}, x0, x1, y0, y1, ex, ey);
                in_y = false;
            }
        }
        const bool more = it < it_end;           // (uniform) next segment: its A loads go out first, the flush rides behind them
        // asm accesses straight behind the last MFMAs: "XDL write VGPR -> VMEM read" (11 wait states for an 8-pass MFMA) and the
        // overwrite of MFMA source registers are software hazards that hipcc's recognizer does not see through inline asm
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        if (F16) {
            // the last item's column exponents: behind their loads went at most 2 (D + SPH) operations (its own pieces and the
            // stores of the item before it); 2 D is a bound that holds for a one-item segment as well
            wait_vmcnt<2 * D>();
            if (in_y) asm volatile("" : "+v"(ey[0]), "+v"(ey[1]));
            else asm volatile("" : "+v"(ex[0]), "+v"(ex[1]));
        }
        if (more) {
            decode(it, b, g, band, c0i);
            issue_a(b, band);
        }
        if (in_y) flush(y0, y1, ey);
        else flush(x0, x1, ex);
        if (!more) break;
        seg_setup();                             // the next segment's first item waits for its A units k-step by k-step
        item(WF0{}, WF1{}, No{}, K1{}, x0, x1, x0, x1, ex, ex);
        row_exponents();
    }
    wait_vmcnt<0>();                             // nothing of this workgroup may still be in flight towards its LDS when it retires
}

static int cu_count() {   // of the CURRENT device (cached per ordinal: a process may drive GPUs of different sizes)
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int n = cus[dev].load(std::memory_order_relaxed);
    if (!n) {
        hipDeviceProp_t prop;
        n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

}  // namespace

#ifdef MV_SPLIT_PROBE
static long long* g_stamp_host = nullptr;
extern "C" int mv_split_probe_stamps(long long* host_out, size_t n) {   // (probe builds only)
    if (!g_stamp_host) return -1;
    return hipMemcpy(host_out, g_stamp_host, n * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif

static int pieces_of(int mode) { return mode == MV_PACK_BF16X3 ? 3 : mode == MV_PACK_F16X2 ? 2 : 0; }

extern "C" size_t mv_volume_pack_bytes(int B, int C, int N, int mode) {
    const int np = pieces_of(mode);
    if (np == 0 || B <= 0 || C <= 0 || (C % 16) || C > 512 || N <= 0) return 0;
    const size_t nrb = (size_t)((N + 31) / 32 + 1);
    const size_t units = (size_t)B * nrb * (size_t)(C / 16) * (size_t)np * 1024;
    const size_t exps = mode == MV_PACK_F16X2 ? (((size_t)B * nrb * 32 * sizeof(int32_t) + 1023) & ~(size_t)1023) : 0;   // per-row scale exponents
    return units + exps;
}

static int volume_pack_impl(const float* f1, const float* f2, void* packed1, void* packed2, int B, int C, int N1, int N2,
                            int layout, int mode, int tile_w, mvStream_t stream) {
    MV_CHECK_ARG(f1 && f2 && packed1 && packed2 && B > 0 && N1 > 0 && N2 > 0);
    MV_CHECK_ARG(layout == MV_LAYOUT_CHW || layout == MV_LAYOUT_HWC);
    MV_CHECK_ARG(((uintptr_t)f1 & 15) == 0 && ((uintptr_t)f2 & 15) == 0 && ((uintptr_t)packed1 & 15) == 0 && ((uintptr_t)packed2 & 15) == 0);
    if (pieces_of(mode) == 0 || C <= 0 || (C % 16) || C > 512) return MV_ERR_UNSUPPORTED;
    const long blocks = (long)B * ((std::max(N1, N2) + 31) / 32 + 1);
    if (blocks > 0x7fffffffL) return MV_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)blocks, 2), blk(256);
    const int hwc = layout == MV_LAYOUT_HWC ? 1 : 0;
    if (mode == MV_PACK_BF16X3)
        hipLaunchKernelGGL((volume_pack_kernel<3, false>), grid, blk, 0, (hipStream_t)stream, f1, f2, (uint16_t*)packed1, (uint16_t*)packed2, B, C,
                           N1, N2, hwc, tile_w);
    else
        hipLaunchKernelGGL((volume_pack_kernel<2, true>), grid, blk, 0, (hipStream_t)stream, f1, f2, (uint16_t*)packed1, (uint16_t*)packed2, B, C,
                           N1, N2, hwc, tile_w);
    return mv_launch_status();
}

extern "C" int mv_volume_pack(const float* f1, const float* f2, void* packed1, void* packed2, int B, int C, int N1, int N2,
                              int layout, int mode, mvStream_t stream) {
    return volume_pack_impl(f1, f2, packed1, packed2, B, C, N1, N2, layout, mode, 0, stream);
}

// ... with operand 2 (an H2 x W2 map, both multiples of 4) in 4 x 4-tile order: mv_corr_volume_packed then writes the TILED volume
// out[b][i][(ty * W2/4 + tx) * 16 + (y % 4) * 4 + x % 4] that mv_corr_lookup_tiled reads (frame driver, lanes >= 3)
extern "C" int mv_volume_pack_tiled(const float* f1, const float* f2, void* packed1, void* packed2, int B, int C, int N1, int H2,
                                    int W2, int layout, int mode, mvStream_t stream) {
    MV_CHECK_ARG(H2 > 0 && W2 > 0 && (H2 % 4) == 0 && (W2 % 4) == 0);
    return volume_pack_impl(f1, f2, packed1, packed2, B, C, N1, H2 * W2, layout, mode, W2, stream);
}

// shapes the streaming GEMM covers (the caller falls back to the exact fp32 kernel otherwise — never less accurate)
extern "C" int mv_corr_volume_packed_supported(int B, int C, int N1, int N2, int mode) {
    // (cu_count() >= 8: the persistent grid is a multiple of 8 workgroups, one run per XCD — part of "supported" so that callers fall back BEFORE they pack)
    return cu_count() >= 8 && pieces_of(mode) != 0 && C == 256 && B > 0 && B <= 65535 && N1 >= 32 && N2 >= 64 && (N2 % 64) == 0 &&
           ((size_t)N1 * N2) < ((size_t)1 << 30) &&                                               // 32-bit byte offsets inside a pair's block
           (size_t)B * (size_t)((N1 + 127) / 128) * (size_t)(N2 / 64) < ((size_t)1 << 31);        // int item index
}

extern "C" int mv_corr_volume_packed(const void* packed1, const void* packed2, float* out, int B, int C, int N1, int N2, int mode,
                                     mvStream_t stream) {
    return mv_corr_volume_packed_shared(packed1, packed2, out, B, C, N1, N2, mode, 0, stream);
}

// ... leaving `free_cus` CUs (rounded down to a multiple of 8: one per XCD and run) without a persistent workgroup.  For a caller that runs other kernels
// BESIDE the GEMM (the frame driver): the f16x2 waves hold 300 of a SIMD's 512 registers, so a workgroup that needs more than the rest (the LM solve: 496, tools/kernel_resources.py) can
// only start on a CU without a GEMM workgroup — with none free it waits for the gap between two GEMMs.  Same bits for any value.
extern "C" int mv_corr_volume_packed_shared(const void* packed1, const void* packed2, float* out, int B, int C, int N1, int N2, int mode,
                                            int free_cus, mvStream_t stream) {
    MV_CHECK_ARG(packed1 && packed2 && out && free_cus >= 0);
    MV_CHECK_ARG(((uintptr_t)packed1 & 15) == 0 && ((uintptr_t)packed2 & 15) == 0);
    if (!mv_corr_volume_packed_supported(B, C, N1, N2, mode)) return MV_ERR_UNSUPPORTED;
    const int np = pieces_of(mode);
    // the dynamic-LDS limit is a per-device function attribute: remembered per device ordinal (a process that drives a second GPU must
    // set it there too); the flag array is only ever written with `true`, so concurrent first calls are benign
    static std::atomic<bool> attr_done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[dev].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute((const void*)corr_volume_split_stream<3, false, 16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)corr_volume_split_stream<2, true, 16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done[dev].store(true, std::memory_order_release);
    }
    if (cu_count() < 8) return MV_ERR_UNSUPPORTED;   // the persistent grid is a multiple of 8 workgroups (one run per XCD): callers fall back to the exact kernel
    // (round-3 / round-4 A/B forms, removed in round 5: the f16x2 kernel as one 8-wave workgroup per CU — no faster alone, 4.40 k vs 4.84 k frames/s in the
    // pipe —; fewer persistent workgroups; workgroups claiming 160 KB of LDS to keep LDS-using kernels off their CUs — both slower.  DESIGN.md changelog.)
    // column regions: one region's B planes (nc / R sub-tiles x 64 rows x C x 2 B x pieces) <= 4 MB.  Measured at 640x480 (7.4 MB
    // of B planes per pair, all of it Infinity-Cache resident): R = 1 / 2 / 3 / 4 / 6 -> 117 / 110 / 113 / 114 / 116 us: fewer, longer
    // band segments (each segment change reloads 192 KB of A fragments per workgroup, ~3 us) against L2 hits on the B sub-tiles
    static int regs_env = -1;   // MV_SPLIT_REGIONS: A/B knob
    if (regs_env < 0) { const char* e = getenv("MV_SPLIT_REGIONS"); regs_env = e ? atoi(e) : 0; }
    const int nc = N2 / 64;
    int R = regs_env > 0 ? regs_env : (int)(((size_t)nc * 64 * C * 2 * np + (4u << 20) - 1) / (4u << 20));
    R = std::max(1, std::min(R, nc));
#ifdef MV_SPLIT_PROBE
    {
        static int dbg = -1;
        if (dbg < 0) {
            const char* e = getenv("MV_SPLIT_DBG");
            dbg = e ? atoi(e) : 0;
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_split_dbg), &dbg, sizeof(int));
            if (dbg & 16) {
                long long* buf = nullptr;
                (void)hipMalloc((void**)&buf, (size_t)2048 * 4 * 64 * 8 * sizeof(long long));
                (void)hipMemset(buf, 0, (size_t)2048 * 4 * 64 * 8 * sizeof(long long));
                (void)hipMemcpyToSymbol(HIP_SYMBOL(g_split_stamps), &buf, sizeof(buf));
                g_stamp_host = buf;
            }
        }
    }
#endif
    // one workgroup per CU, a multiple of 8 (one run per XCD), minus the CUs the caller keeps free
    const int full = cu_count() & ~7, keep = free_cus & ~7;
    const dim3 g(full - keep >= 8 ? full - keep : 8);
    auto lds_bytes = [&](size_t need) { return need; };
    if (mode == MV_PACK_BF16X3) {
        using K = SplitCfg<3, false, 16, 4>;
        mv_note_volume_kernel("corr_volume_split_stream<bf16x3>");
        hipLaunchKernelGGL((corr_volume_split_stream<3, false, 16, 4>), g, dim3(256), lds_bytes(K::NSLOT * K::SLOT_BYTES), (hipStream_t)stream,
                           (const uint16_t*)packed1, (const uint16_t*)packed2, out, N1, N2, B, R);
    } else {
        using K = SplitCfg<2, true, 16, 4>;
        mv_note_volume_kernel("corr_volume_split_stream<f16x2>");
        hipLaunchKernelGGL((corr_volume_split_stream<2, true, 16, 4>), g, dim3(256), lds_bytes(K::NSLOT * K::SLOT_BYTES), (hipStream_t)stream,
                           (const uint16_t*)packed1, (const uint16_t*)packed2, out, N1, N2, B, R);
    }
    return mv_launch_status();
}
