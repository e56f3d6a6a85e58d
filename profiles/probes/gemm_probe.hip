// scratch: which part of the fp32 volume GEMM costs the 33 % between 105 and 156 TFLOP/s?  (knock-out variants)
#define MV_VOL_RG_RUNTIME 1
#include "../../mac-vo_amd/csrc/corr_volume.hip"
namespace {
template <int MODE>   // 1 = no epilogue store, 2 = no global loads, 4 = no ds_write + barrier, 8 = no LDS fragment reads
__global__ __launch_bounds__(256) void probe_gemm(const float* __restrict__ f1, const float* __restrict__ f2,
                                                  float* __restrict__ out, int C, int N1, int N2, int tiles_m, int tiles_n) {
    constexpr int BK = 16, NP = BK / 8;
    __shared__ __attribute__((aligned(16))) float sA[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float sB[2][BK][BN];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1, kh = lane >> 5, li = lane & 31;
    const int lrow = t >> 5, lcol = (t & 31) * 4, nk = C / BK;
    int tm, tn;
    const int b = blockIdx.z;
    tile_coords(tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const float* A = f1 + (size_t)b * C * N1;
    const float* Bp = f2 + (size_t)b * C * N2;
    f32x4 ra[NP], rb[NP];
    for (int p = 0; p < NP; ++p) { ra[p] = f32x4{1, 2, 3, 4}; rb[p] = f32x4{4, 3, 2, 1}; }
    auto gload = [&](int k0) {
        if (MODE & 2) return;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int k = k0 + lrow + 8 * p;
            const float* pa = A + (size_t)k * N1 + m0 + lcol;
            const float* pb = Bp + (size_t)k * N2 + n0 + lcol;
            ra[p] = (m0 + lcol < N1) ? *reinterpret_cast<const f32x4*>(pa) : f32x4{0, 0, 0, 0};
            rb[p] = (n0 + lcol < N2) ? *reinterpret_cast<const f32x4*>(pb) : f32x4{0, 0, 0, 0};
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            *reinterpret_cast<f32x4*>(&sA[buf][lrow + 8 * p][lcol]) = ra[p];
            *reinterpret_cast<f32x4*>(&sB[buf][lrow + 8 * p][lcol]) = rb[p];
        }
    };
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
        if (MODE & 8) {
            float x = (float)lane, y = (float)wave;
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i][j], 0, 0, 0);
            }
        } else {
            mfma_tile_f32<BK, BM>(sA[buf], sB[buf], wm * 64 + li, wn * 64 + li, kh, acc);
        }
        if (kt + 1 < nk && !(MODE & 4)) {
            sstore(buf ^ 1);
            __syncthreads();
        }
    }
    if (MODE & 1) {
        float s = 0.f;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        if (s == 123.456f) out[0] = s;
    } else {
        const bool interior = (m0 + BM <= N1) && (n0 + BN <= N2);
        store_tile(out + (size_t)b * N1 * N2, acc, m0 + wm * 64, n0 + wn * 64, kh, li, N1, N2, interior);
    }
}
}  // namespace
#include <type_traits>

namespace {
// 2-deep global prefetch, register staged: loads of tile kt+2 are issued while tile kt is multiplied; the wait before the
// LDS store of tile kt+1 is vmcnt(4) (the 4 younger loads stay in flight) instead of vmcnt(0)
__constant__ int g_stagger = 4;
template <int MODE>
__global__ __launch_bounds__(256) void probe_gemm_pf2(const float* __restrict__ f1, const float* __restrict__ f2,
                                                      float* __restrict__ out, int C, int N1, int N2, int tiles_m, int tiles_n) {
    constexpr int BK = 16, NP = BK / 8;
    __shared__ __attribute__((aligned(16))) float sA[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float sB[2][BK][BN];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1, kh = lane >> 5, li = lane & 31;
    const int lrow = t >> 5, lcol = (t & 31) * 4, nk = C / BK;
    int tm, tn;
    const int b = blockIdx.z;
    tile_coords(tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    // columns past the edge are clamped to the last valid float4: they only feed output rows / columns that are never stored
    const float* A = f1 + (size_t)b * C * N1 + min(m0 + lcol, N1 - 4);
    const float* Bp = f2 + (size_t)b * C * N2 + min(n0 + lcol, N2 - 4);
    f32x4 ra[2][NP], rb[2][NP];
    auto gload = [&](auto SET, int k0) {
        constexpr int S = decltype(SET)::value;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int k = k0 + lrow + 8 * p;
            ra[S][p] = *reinterpret_cast<const f32x4*>(A + (size_t)k * N1);
            rb[S][p] = *reinterpret_cast<const f32x4*>(Bp + (size_t)k * N2);
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
    if (MODE & 2) {   // experiment: phase-shift the first wave of workgroups so that tile completions (store bursts) spread out
        const int lin = blockIdx.z * gridDim.x + blockIdx.x;
        if (lin < 1024) {
            const int q = lin >> 8;                      // 0..3: presumably the slot index on its CU
            for (int i = 0; i < q * g_stagger; ++i) __builtin_amdgcn_s_sleep(127);   // 127 * 64 clk = 3.4 us each
        }
    }
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // nk is even (C % 32 == 0).  Loads are issued UNCONDITIONALLY (the last two re-read the final tile): with no branch
    // around them the compiler can count them, and the wait in front of each LDS store becomes vmcnt(4), not vmcnt(0)
    const int klast = C - BK;
    gload(S0{}, 0);
    gload(S1{}, BK);
    sstore(S0{}, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        // even step: tile kt in buf 0, tile kt+1 in regs set 1
        gload(S0{}, min((kt + 2) * BK, klast));
        mfma_tile_f32<BK, BM>(sA[0], sB[0], wm * 64 + li, wn * 64 + li, kh, acc);
        sstore(S1{}, 1);
        __syncthreads();
        // odd step: tile kt+1 in buf 1, tile kt+2 in regs set 0
        gload(S1{}, min((kt + 3) * BK, klast));
        mfma_tile_f32<BK, BM>(sA[1], sB[1], wm * 64 + li, wn * 64 + li, kh, acc);
        sstore(S0{}, 0);
        __syncthreads();
    }
    if (MODE & 1) {
        float s = 0.f;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        if (s == 123.456f) out[0] = s;
    } else if (MODE & 4) {
        // same store instructions, but every workgroup writes into its XCD-local 64 KB slot of a 2 MB window: stays in L2
        float* O = out + (size_t)(blockIdx.x & 31) * 16384;
        store_tile<false>(O, acc, wm * 64, wn * 64, kh, li, 128, 128, true);
    } else if (MODE & 8) {
        const bool interior = (m0 + BM <= N1) && (n0 + BN <= N2);
        store_tile<false>(out + (size_t)b * N1 * N2, acc, m0 + wm * 64, n0 + wn * 64, kh, li, N1, N2, interior);
    } else {
        const bool interior = (m0 + BM <= N1) && (n0 + BN <= N2);
        store_tile(out + (size_t)b * N1 * N2, acc, m0 + wm * 64, n0 + wn * 64, kh, li, N1, N2, interior);
    }
}
}  // namespace

namespace {
// direct-to-LDS (global_load_lds_dwordx4), NS LDS stages, counted vmcnt: tile kt+NS-1 is issued while tile kt is multiplied
template <int NS, bool PRIO>
__global__ __launch_bounds__(256) void probe_gemm_glds(const float* __restrict__ f1, const float* __restrict__ f2,
                                                       float* __restrict__ out, int C, int N1, int N2, int tiles_m, int tiles_n) {
    constexpr int BK = 16;
    __shared__ __attribute__((aligned(1024))) float sA[NS][BK][BM];
    __shared__ __attribute__((aligned(1024))) float sB[NS][BK][BN];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1, kh = lane >> 5, li = lane & 31;
    const int lrow = t >> 5, lcol = (t & 31) * 4, nk = C / BK;
    int tm, tn;
    const int b = blockIdx.z;
    tile_coords(tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const float* A = f1 + (size_t)b * C * N1 + min(m0 + lcol, N1 - 4);
    const float* Bp = f2 + (size_t)b * C * N2 + min(n0 + lcol, N2 - 4);
    const int klast = C - BK;
    // a wave's 64 lanes x 16 B land at (wave-uniform LDS base) + lane * 16: rows 2*wave, 2*wave+1 of the [BK][128] tile
    auto issue = [&](int stage, int k0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int k = k0 + lrow + 8 * p;
            unsigned keep;
            const unsigned da = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&sA[stage][2 * wave + 8 * p][0]);
            const unsigned db = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&sB[stage][2 * wave + 8 * p][0]);
            const float* ga = A + (size_t)k * N1;
            const float* gb = Bp + (size_t)k * N2;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(ga), "s"(da) : "memory");
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gb), "s"(db) : "memory");
        }
    };
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s, min(s * BK, klast));
    if (NS == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int stage = 0, nstage = NS - 1;
    for (int kt = 0; kt < nk; ++kt) {
        issue(nstage, min((kt + NS - 1) * BK, klast));
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        mfma_tile_f32<BK, BM>(sA[stage], sB[stage], wm * 64 + li, wn * 64 + li, kh, acc);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        // tile kt+1 must have landed (everything but the youngest NS-2 tiles' pieces), for every wave
        if (NS == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage = stage + 1 == NS ? 0 : stage + 1;
        nstage = nstage + 1 == NS ? 0 : nstage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const bool interior = (m0 + BM <= N1) && (n0 + BN <= N2);
    store_tile(out + (size_t)b * N1 * N2, acc, m0 + wm * 64, n0 + wn * 64, kh, li, N1, N2, interior);
}
}  // namespace

namespace {
// 160 x 128 tile (4800 = 30 x 160: 2280 tiles = 8.9 per CU -> 9, 99 % balanced, vs 11.28 -> 12, 94 %, for 128 x 128):
// 4 waves side by side along N, each 160 x 32 = 5 MFMA blocks; 2-deep register prefetch as the shipped kernel
constexpr int BM2 = 160;
template <int MODE>
__global__ __launch_bounds__(256, 3) void probe_gemm_160(const float* __restrict__ f1, const float* __restrict__ f2,
                                                      float* __restrict__ out, int C, int N1, int N2, int tiles_m, int tiles_n) {
    constexpr int BK = 16;
    __shared__ __attribute__((aligned(16))) float sA[2][BK][BM2];
    __shared__ __attribute__((aligned(16))) float sB[2][BK][BN];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, kh = lane >> 5, li = lane & 31;
    const int nk = C / BK;
    int tm, tn;
    const int b = blockIdx.z;
    tile_coords(tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM2, n0 = tn * BN;
    // loader: A tile = 16 rows x 40 float4 (640), B tile = 16 rows x 32 float4 (512)
    const int arow[3] = {t / 40, (t + 256) / 40, (t + 512) / 40};
    const int acol[3] = {(t % 40) * 4, ((t + 256) % 40) * 4, ((t + 512) % 40) * 4};
    const bool a3 = t + 512 < 640;
    const int brow = t >> 5, bcol = (t & 31) * 4;
    const float* Ab = f1 + (size_t)b * C * N1;
    const float* Bb = f2 + (size_t)b * C * N2 + min(n0 + bcol, N2 - 4);
    const float* pa[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) pa[j] = Ab + (size_t)(j == 2 && !a3 ? 0 : arow[j]) * N1 + min(m0 + acol[j], N1 - 4);
    const int klast = C - BK;
    f32x4 ra[2][3], rb[2][2];
    auto gload = [&](auto SET, int k0) {
        constexpr int S = decltype(SET)::value;
#pragma unroll
        for (int j = 0; j < 3; ++j) ra[S][j] = *reinterpret_cast<const f32x4*>(pa[j] + (size_t)k0 * N1);
#pragma unroll
        for (int p = 0; p < 2; ++p) rb[S][p] = *reinterpret_cast<const f32x4*>(Bb + (size_t)(k0 + brow + 8 * p) * N2);
    };
    auto sstore = [&](auto SET, int buf) {
        constexpr int S = decltype(SET)::value;
        *reinterpret_cast<f32x4*>(&sA[buf][arow[0]][acol[0]]) = ra[S][0];
        *reinterpret_cast<f32x4*>(&sA[buf][arow[1]][acol[1]]) = ra[S][1];
        if (a3) *reinterpret_cast<f32x4*>(&sA[buf][arow[2]][acol[2]]) = ra[S][2];
#pragma unroll
        for (int p = 0; p < 2; ++p) *reinterpret_cast<f32x4*>(&sB[buf][brow + 8 * p][bcol]) = rb[S][p];
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    f32x16 acc[5];
    for (int i = 0; i < 5; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    auto mma = [&](int buf) {
        float a[2][5], bb[2];
#pragma unroll
        for (int i = 0; i < 5; ++i) a[0][i] = sA[buf][kh][i * 32 + li];
        bb[0] = sB[buf][kh][wave * 32 + li];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
            if (kk + 2 < BK) {
#pragma unroll
                for (int i = 0; i < 5; ++i) a[nxt][i] = sA[buf][kk + 2 + kh][i * 32 + li];
                bb[nxt] = sB[buf][kk + 2 + kh][wave * 32 + li];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 5; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], bb[cur], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    gload(S0{}, 0);
    gload(S1{}, BK);
    sstore(S0{}, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        gload(S0{}, min((kt + 2) * BK, klast));
        mma(0);
        sstore(S1{}, 1);
        __syncthreads();
        gload(S1{}, min((kt + 3) * BK, klast));
        mma(1);
        sstore(S0{}, 0);
        __syncthreads();
    }
    if (MODE & 1) {
        float s = 0.f;
        for (int i = 0; i < 5; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        if (s == 123.456f) out[0] = s;
    } else {
        float* O = out + (size_t)b * N1 * N2;
        const int col = n0 + wave * 32 + li;
        if (m0 + BM2 <= N1 && n0 + BN <= N2) {   // interior tile: no per-element guards
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_nontemporal_store(acc[i][r], O + (size_t)(m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * N2 + col);
        } else {
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (row < N1 && col < N2) __builtin_nontemporal_store(acc[i][r], &O[(size_t)row * N2 + col]);
                }
        }
    }
}
}  // namespace

extern "C" int gemm_probe_set_stagger(int v) { return hipMemcpyToSymbol(HIP_SYMBOL(g_stagger), &v, sizeof(int)) == hipSuccess ? 0 : 1; }
extern "C" int gemm_probe_set_rg(int rg) { return hipMemcpyToSymbol(HIP_SYMBOL(g_rg), &rg, sizeof(int)) == hipSuccess ? 0 : 1; }
extern "C" int gemm_probe_launch(const float* f1, const float* f2, float* out, int B, int C, int N, int mode, void* stream) {
    const int tm = (N + BM - 1) / BM;
    dim3 grid(tm * tm, 1, B), block(256);
    hipStream_t s = (hipStream_t)stream;
#define L(M) case M: hipLaunchKernelGGL(probe_gemm<M>, grid, block, 0, s, f1, f2, out, C, N, N, tm, tm); break
    if (mode == 30 || mode == 31) {
        const int tm2 = (N + BM2 - 1) / BM2;
        dim3 g2(tm2 * tm, 1, B);
        if (mode == 30) hipLaunchKernelGGL(probe_gemm_160<0>, g2, block, 0, s, f1, f2, out, C, N, N, tm2, tm);
        else hipLaunchKernelGGL(probe_gemm_160<1>, g2, block, 0, s, f1, f2, out, C, N, N, tm2, tm);
        return hipGetLastError() == hipSuccess ? 0 : 1;
    }
    if (mode == 20) { hipLaunchKernelGGL((probe_gemm_glds<3, false>), grid, block, 0, s, f1, f2, out, C, N, N, tm, tm); return hipGetLastError() == hipSuccess ? 0 : 1; }
    if (mode == 21) { hipLaunchKernelGGL((probe_gemm_glds<3, true>), grid, block, 0, s, f1, f2, out, C, N, N, tm, tm); return hipGetLastError() == hipSuccess ? 0 : 1; }
    if (mode == 22) { hipLaunchKernelGGL((probe_gemm_glds<4, false>), grid, block, 0, s, f1, f2, out, C, N, N, tm, tm); return hipGetLastError() == hipSuccess ? 0 : 1; }
    if (mode == 40) { hipLaunchKernelGGL(probe_gemm_pf2<4>, grid, block, 0, s, f1, f2, out, C, N, N, tm, tm); return hipGetLastError() == hipSuccess ? 0 : 1; }
    if (mode == 41) { hipLaunchKernelGGL(probe_gemm_pf2<8>, grid, block, 0, s, f1, f2, out, C, N, N, tm, tm); return hipGetLastError() == hipSuccess ? 0 : 1; }
    if (mode == 18) { hipLaunchKernelGGL(probe_gemm_pf2<2>, grid, block, 0, s, f1, f2, out, C, N, N, tm, tm); return hipGetLastError() == hipSuccess ? 0 : 1; }
    if (mode == 16) { hipLaunchKernelGGL(probe_gemm_pf2<0>, grid, block, 0, s, f1, f2, out, C, N, N, tm, tm); return hipGetLastError() == hipSuccess ? 0 : 1; }
    if (mode == 17) { hipLaunchKernelGGL(probe_gemm_pf2<1>, grid, block, 0, s, f1, f2, out, C, N, N, tm, tm); return hipGetLastError() == hipSuccess ? 0 : 1; }
    switch (mode) { L(0); L(1); L(2); L(3); L(4); L(5); L(6); L(7); L(8); L(9); L(15); L(14); L(12); L(10); L(11); L(13); default: return 2; }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
