#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <type_traits>
#define MV_VOL_STORE_ASM_MOD ""
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait for this wave's older memory operations (all but the newest N), then the workgroup barrier — one statement so that
// nothing can be scheduled between the two
template <int N>
__device__ __forceinline__ void wait_vmcnt_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
// LDS-DMA: 64 lanes x 16 B from per-lane global addresses to LDS bytes [lds_dst, lds_dst + 1024) in lane order.  M0 carries the
// destination and is compiler-reserved: saved, written and restored inside the one statement.  Invisible to hipcc's s_waitcnt
// bookkeeping (counted by hand, see the kernel).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

template <bool IS_BF16, int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void corr_volume_h_stream(
    const uint16_t* __restrict__ f1, const uint16_t* __restrict__ f2, float* __restrict__ out, int N1, int N2, int B, int R, long long* __restrict__ ts) {
    int tsn = 0;
    const bool rec = (threadIdx.x == 0) && (blockIdx.x % 37 == 0);
    long long* tsp = ts + (blockIdx.x / 37) * 256;
#define STAMP() do { if (rec && tsn < 255) tsp[tsn++] = clock64(); } while (0)
    STAMP();
    constexpr int C = KS * 16, CH = KS * 2;      // channels; 16-byte chunks per row
    constexpr int RPI = 64 / CH;                 // B rows one wave-wide LDS-DMA instruction covers (64 lanes x 16 B = 1 KB)
    constexpr int NP = 64 / (4 * RPI);           // LDS-DMA instructions per wave and sub-tile (4 waves)
    constexpr int RPK = 16 / KS;                 // accumulator rows stored per k-step of the NEXT sub-tile
    constexpr int SLOT = 64 * CH;                // ring slot in 16-byte units
    static_assert(NP <= 8 && RPK * KS == 16, "the vmcnt budget below assumes <= 8 loads and 32 stores per sub-tile");
    extern __shared__ __attribute__((aligned(16))) i32x4 smem_hs[];   // B ring: 2 slots x 64 rows x CH chunks
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
        lane_src[p] = (unsigned)(row * C + ((pos ^ (row & 15)) << 3));
    }
    auto issue_b = [&](int slot) __attribute__((always_inline)) {   // item ld_it -> ring slot, then advance
#pragma unroll
        for (int p = 0; p < NP; ++p) glds16(ld_ptr + lane_src[p], lds0 + (unsigned)(slot * SLOT + (p * 4 + wave) * 64) * 16u);
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
    issue_b(0);
    int slot = 0;
    const int sw = li & 15;                      // fragment rows li and li + 32 share the swizzle
    float* O = nullptr;                          // wave-uniform: column 0 of the CURRENT item's output block (row 0 of the pair)
    // per-lane BYTE offsets of the 16 accumulator rows (C/D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)),
    // fixed for a band: stores are `uniform base + 32-bit lane offset`, no address arithmetic between the MFMAs.
    // Rows past N1 (last band) hold copies of row N1 - 1 (the A rows are clamped the same way): they are stored ON TOP of row
    // N1 - 1 with identical values instead of being branched around.
    unsigned roff[16];
    auto store_r = [&](const f32x16& p0, const f32x16& p1, int r, float* Ob) __attribute__((always_inline)) {
        // asm: hipcc strength-reduces `Ob + roff[r]` into sixteen 64-bit per-lane pointers (32 VGPRs -> spills at the 256-register
        // budget of 2 waves / SIMD); the SGPR-base form needs none.  Like the DMA above these are counted by hand.
        asm volatile("global_store_dword %0, %1, %2" MV_VOL_STORE_ASM_MOD ::"v"(roff[r]), "v"(p0[r]), "s"(Ob) : "memory");
        asm volatile("global_store_dword %0, %1, %2 offset:128" MV_VOL_STORE_ASM_MOD ::"v"(roff[r]), "v"(p1[r]), "s"(Ob) : "memory");
    };
    // one sub-tile: ring upkeep, 2 x KS MFMAs into (c0, c1); with PREV the 32 stores of (p0, p1) ride between them
    auto step = [&](auto PREV, f32x16& c0, f32x16& c1, const f32x16& p0, const f32x16& p1) __attribute__((always_inline)) {
        constexpr bool HAVE_PREV = decltype(PREV)::value;
        // this wave's DMA pieces of ring slot `slot` were issued before the last 32 stores (or everything has been drained since);
        // behind the barrier all four waves' pieces are in, and everyone has left slot ^ 1
        STAMP();
        wait_vmcnt_barrier<32>();
        STAMP();
        issue_b(slot ^ 1);
        STAMP();
#pragma unroll
        for (int r = 0; r < 16; ++r) c0[r] = c1[r] = 0.f;
        const i32x4* q0 = smem_hs + slot * SLOT + li * CH;
        const i32x4* q1 = q0 + 32 * CH;
        constexpr int PF = 2;                    // B fragments are fetched PF k-steps ahead of the MFMAs that use them
        i32x4 fb0[PF + 1], fb1[PF + 1];
#pragma unroll
        for (int ks = 0; ks < PF; ++ks) {
            fb0[ks] = q0[(ks * 2 + kh) ^ sw];
            fb1[ks] = q1[(ks * 2 + kh) ^ sw];
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + PF < KS) {
                fb0[(ks + PF) % (PF + 1)] = q0[((ks + PF) * 2 + kh) ^ sw];
                fb1[(ks + PF) % (PF + 1)] = q1[((ks + PF) * 2 + kh) ^ sw];
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of the MFMAs (hipcc otherwise sinks it back)
            const i32x4 b0 = fb0[ks % (PF + 1)], b1 = fb1[ks % (PF + 1)];
            if (IS_BF16) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[ks]), __builtin_bit_cast(bf16x8, b0), c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[ks]), __builtin_bit_cast(bf16x8, b1), c1, 0, 0, 0);
            } else {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[ks]), __builtin_bit_cast(f16x8, b0), c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[ks]), __builtin_bit_cast(f16x8, b1), c1, 0, 0, 0);
            }
            if (HAVE_PREV) {
#pragma unroll
                for (int q = 0; q < RPK; ++q) store_r(p0, p1, ks * RPK + q, O - 64);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP();
        if (!HAVE_PREV) wait_vmcnt<0>();         // no stores went out behind this pass's DMA: the next pass's vmcnt(32) would not cover it
        ++it;
        slot ^= 1;
        O += 64;
    };
    auto flush = [&](const f32x16& p0, const f32x16& p1) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) store_r(p0, p1, r, O - 64);
    };
    using Yes = std::true_type;
    using No = std::false_type;
    // whole-K A fragments of this wave's 32 rows of (pair b, band) -> registers.  asm like the DMA: the wait is placed by hand so
    // that the previous band's last 32 stores can be issued BEHIND these loads and drain while the first sub-tile is multiplied.
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
    STAMP();
    while (true) {                               // one pass per (pair, region, band) segment of the run
        const int seg_end = min(it_end, it + (reg_c0(g + 1) - c0i));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            asm volatile("" : "+v"(afr[ks]));    // the fragments count as defined only here, behind the hand-placed wait
            af[ks] = __builtin_bit_cast(s16x8, afr[ks]);
        }
        O = out + (size_t)b * N1 * N2 + (size_t)c0i * 64;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            roff[r] = ((unsigned)min(band * 128 + wave * 32 + 4 * kh + (r & 3) + 8 * (r >> 2), N1 - 1) * (unsigned)N2 + li) * 4u;
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
        if (!more) { STAMP(); wait_vmcnt<0>(); STAMP(); if (rec) { tsp[254] = wall_clock64(); tsp[255] = tsn; } break; }
        STAMP();
        wait_vmcnt<32>();
        STAMP();                        // the A loads precede the 32 flush stores
    }
}


extern "C" int run(const void* a, const void* b, float* out, int N1, int N2, int B, int R, int grid, int lds, hipStream_t s, long long* ts) {
    static bool done = false; if (!done) { hipFuncSetAttribute((const void*)corr_volume_h_stream<false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); done = true; }
    hipLaunchKernelGGL((corr_volume_h_stream<false, 16>), dim3(grid), dim3(256), lds, s, (const uint16_t*)a, (const uint16_t*)b, out, N1, N2, B, R, ts);
    return (int)hipGetLastError();
}
