// scratch (round 2): fp32 volume GEMM variants.  Question: the shipped kernel allocates 148 registers (84 VGPR, 32 of them
// hoisted LDS row addresses for ds_read2_b32, whose 8-bit offsets cannot reach row k + 2) -> 3 waves / SIMD.  Does a layout
// whose fragment reads are ONE base register + immediates (ds_read2st64_b32) and the 4th wave per SIMD help?
#include "../../mac-vo_amd/csrc/corr_volume.hip"
namespace {

// LDS tile [BK][128] with permuted columns: col = w*64 + i*32 + l  ->  p = i*64 + w*32 + l, so a wave's two fragments of one
// k row are 64 floats (256 B) apart and consecutive k rows 512 B: every fragment address = base + immediate.
__device__ __forceinline__ int perm_col(int col) { return ((col >> 5) & 1) * 64 + (col >> 6) * 32 + (col & 31); }

template <int BK>
__device__ __forceinline__ void mfma_tile_perm(const float* __restrict__ pa, const float* __restrict__ pb, f32x16 (&acc)[2][2]) {
    // pa / pb already include stage, kh row and the wave / lane column; rows are 128 floats apart, k-pairs 256
    float a[2][2], b[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { a[0][i] = pa[i * 64]; b[0][i] = pb[i * 64]; }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
        const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
        if (kk + 2 < BK) {
#pragma unroll
            for (int i = 0; i < 2; ++i) { a[nxt][i] = pa[(kk + 2) * 128 + i * 64]; b[nxt][i] = pb[(kk + 2) * 128 + i * 64]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int WPS>   // launch-bounds hint: waves per SIMD
__global__ __launch_bounds__(256, WPS) void gemm_perm(const float* __restrict__ f1, const float* __restrict__ f2,
                                                       float* __restrict__ out, int C, int N1, int N2, int tiles_m, int tiles_n) {
    constexpr int BK = 16, NP = BK / 8;
    // one array: [stage][operand][BK][128]
    __shared__ __attribute__((aligned(16))) float smem[2 * 2 * BK * 128];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1, kh = lane >> 5, li = lane & 31;
    const int lrow = t >> 5, lcol = (t & 31) * 4, nk = C / BK;
    int tm, tn;
    const int b = blockIdx.z;
    tile_coords(tiles_m, tiles_n, tm, tn);
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
    float* wbase = smem + lrow * 128 + perm_col(lcol);   // + stage*2*BK*128 + operand*BK*128 + p*8*128
    auto sstore = [&](auto SET, int buf) {
        constexpr int S = decltype(SET)::value;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            *reinterpret_cast<f32x4*>(wbase + buf * 2 * BK * 128 + p * 8 * 128) = ra[S][p];
            *reinterpret_cast<f32x4*>(wbase + buf * 2 * BK * 128 + BK * 128 + p * 8 * 128) = rb[S][p];
        }
    };
    const float* fa = smem + kh * 128 + wm * 32 + li;
    const float* fb = smem + BK * 128 + kh * 128 + wn * 32 + li;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    gload(S0{}, 0);
    gload(S1{}, BK);
    sstore(S0{}, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        gload(S0{}, min((kt + 2) * BK, klast));
        mfma_tile_perm<BK>(fa, fb, acc);
        sstore(S1{}, 1);
        __syncthreads();
        gload(S1{}, min((kt + 3) * BK, klast));
        mfma_tile_perm<BK>(fa + 2 * BK * 128, fb + 2 * BK * 128, acc);
        sstore(S0{}, 0);
        __syncthreads();
    }
    const bool interior = (m0 + BM <= N1) && (n0 + BN <= N2);
    store_tile(out + (size_t)b * N1 * N2, acc, m0 + wm * 64, n0 + wn * 64, kh, li, N1, N2, interior);
}

// ------------------------------------------------------------------------------------------------------------------
// Mixed-tile persistent schedule for N = 4800, B = 2 on 256 CUs x 4 workgroups: every workgroup computes exactly
// 2 tiles of 128x128, 1 of 128x64 and 1 of 64x64 (44 of the 45000/1024 = 43.95 MFMA blocks): no tail, no dead rows.
//   per pair: 128x128 tiles cover [0,4096)^2 (32 x 32); 128x64 tiles cover rows [0,4096) x cols [4096,4800) (32 x 11) and rows
//   [4096,4736) x cols [0,2048) (5 x 32); 64x64 tiles cover rows [4096,4736) x cols [2048,4800) (10 x 43) and rows [4736,4800) (75)
template <int MI, int NJ, bool PTRINC, int BK = 16, int OPT = 0>
__device__ __forceinline__ void tile_body(const float* __restrict__ A, const float* __restrict__ Bp, float* __restrict__ O, int C,
                                          int N1, int N2, int m0, int n0, float* smem) {
    constexpr int WA = 64 * MI, WB = 64 * NJ;             // operand tile widths (floats per k row)
    constexpr int TA = WA / 4, TB = WB / 4;               // threads per k row
    constexpr int NPA = BK * TA / 256, NPB = BK * TB / 256;   // float4 per thread per stage (2 or 1)
    constexpr int STAGE = BK * (WA + WB);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1, kh = lane >> 5, li = lane & 31;
    const int nk = C / BK, klast = C - BK;
    // loader mapping per operand
    const int arow = t / TA, acol = (t % TA) * 4, brow = t / TB, bcol = (t % TB) * 4;
    // uniform (SGPR) bases + ONE 32-bit per-lane offset per operand: global_load ... saddr form, no 64-bit address VGPRs
    const float* Au = A + m0;
    const float* Bu = Bp + n0;
    const unsigned aoff = arow * N1 + acol, boff = brow * N2 + bcol;
    // LDS column permutation: tile column = w * (32 * M) + i * 32 + l (wave row / column w, fragment i)  ->  i * 64 + w * 32 + l:
    // a wave's fragments of one k row are 64 floats (256 B) apart => every fragment read is base + immediate (ds_read2st64_b32)
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
        for (int p = 0; p < NPA; ++p) ra[S][p] = *reinterpret_cast<const f32x4*>(Au + (size_t)(k0 + (256 / TA) * p) * N1 + aoff);
#pragma unroll
        for (int p = 0; p < NPB; ++p) rb[S][p] = *reinterpret_cast<const f32x4*>(Bu + (size_t)(k0 + (256 / TB) * p) * N2 + boff);
    };
    auto sstore = [&](auto SET, int buf) {
        constexpr int S = decltype(SET)::value;
#pragma unroll
        for (int p = 0; p < NPA; ++p) *reinterpret_cast<f32x4*>(wa + buf * STAGE + p * (256 / TA) * WA) = ra[S][p];
#pragma unroll
        for (int p = 0; p < NPB; ++p) *reinterpret_cast<f32x4*>(wb + buf * STAGE + p * (256 / TB) * WB) = rb[S][p];
    };
    // one piece (float4) of the register-staged tile -> LDS; pieces 0..NPA-1 = A, then B
    auto spiece = [&](auto SET, int buf, int g) {
        constexpr int S = decltype(SET)::value;
#pragma unroll
        for (int p = 0; p < NPA; ++p)
            if (g == p) *reinterpret_cast<f32x4*>(wa + buf * STAGE + p * (256 / TA) * WA) = ra[S][p];
#pragma unroll
        for (int p = 0; p < NPB; ++p)
            if (g == NPA + p) *reinterpret_cast<f32x4*>(wb + buf * STAGE + p * (256 / TB) * WB) = rb[S][p];
    };
    constexpr bool SPREAD = (OPT & 1) != 0;
    constexpr int PF = (OPT & 2) ? 2 : 1;          // fragment prefetch distance in k-pairs
    auto mma = [&](int buf, auto&& after) {
        const float* qa = fa + buf * STAGE;
        const float* qb = fb + buf * STAGE;
        float a[PF + 1][MI], b[PF + 1][NJ];
#pragma unroll
        for (int d = 0; d < PF; ++d) {
#pragma unroll
            for (int i = 0; i < MI; ++i) a[d][i] = qa[(2 * d) * WA + i * 64];
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[d][j] = qb[(2 * d) * WB + j * 64];
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int g = kk >> 1;
            const int cur = g % (PF + 1), nxt = (g + PF) % (PF + 1);
            if (kk + 2 * PF < BK) {
#pragma unroll
                for (int i = 0; i < MI; ++i) a[nxt][i] = qa[(kk + 2 * PF) * WA + i * 64];
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[nxt][j] = qb[(kk + 2 * PF) * WB + j * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            after(g);
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    gload(S0{}, 0);
    gload(S1{}, BK);
    __syncthreads();   // the previous tile's last K step may still be reading the LDS stages
    sstore(S0{}, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        gload(S0{}, min((kt + 2) * BK, klast));
        if (SPREAD) {
            mma(0, [&](int g) { spiece(S1{}, 1, g); });
        } else {
            mma(0, [](int) {});
            sstore(S1{}, 1);
        }
        __syncthreads();
        gload(S1{}, min((kt + 3) * BK, klast));
        if (SPREAD) {
            mma(1, [&](int g) { spiece(S0{}, 0, g); });
        } else {
            mma(1, [](int) {});
            sstore(S0{}, 0);
        }
        __syncthreads();
    }
    // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5); wave (wm, wn) owns rows wm*32*MI.., cols wn*32*NJ..
    // NOTE the column permutation only concerns the LDS image; wave wm's fragment i covers tile rows wm*(32*MI) + i*32 .. for MI = 2
    // (w*64 + i*32), and wm*32 for MI = 1
    float* Ou = O + (size_t)m0 * N2 + n0;
    const unsigned so = (wm * 32 * MI + 4 * kh) * N2 + wn * 32 * NJ + li;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float* p = Ou + (size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * N2;
#pragma unroll
            for (int j = 0; j < NJ; ++j) __builtin_nontemporal_store(acc[i][j][r], p + j * 32 + so);
        }
}



// uniform persistent loop over tiles of (64 MI) x (64 NJ): shape exploration at sizes that are multiples of the tile
template <int MI, int NJ, int WPS, int BK = 16, int OPT = 0>
__global__ __launch_bounds__(256, WPS) void gemm_uni(const float* __restrict__ f1, const float* __restrict__ f2, float* __restrict__ out,
                                                      int C, int N, int B) {
    extern __shared__ __attribute__((aligned(16))) float smem_dyn[];
    const int slots = gridDim.x, s = blockIdx.x;
    const size_t fsz = (size_t)C * N, osz = (size_t)N * N;
    const int gm = N / (64 * MI), gn = N / (64 * NJ), per = gm * gn, total = per * B;
    const int lin = (s & 7) * (slots >> 3) + (s >> 3);
#pragma unroll 1
    for (int idx = lin; idx < total; idx += slots) {
        const int b = idx / per, q = idx - b * per;
        const int tm = q / gn, tn = q - tm * gn;
        tile_body<MI, NJ, false, BK, OPT>(f1 + b * fsz, f2 + b * fsz, out + b * osz, C, N, N, tm * 64 * MI, tn * 64 * NJ, smem_dyn);
    }
}
template <int MI, int NJ, int WPS, int BK = 16, int OPT = 0>
static int launch_uni(const float* f1, const float* f2, float* out, int B, int C, int N, int slots, hipStream_t s) {
    if (N % (64 * MI) || N % (64 * NJ) || C % (2 * BK)) return -2;
    const size_t lds = 2 * BK * 64 * (MI + NJ) * sizeof(float);
    auto k = gemm_uni<MI, NJ, WPS, BK, OPT>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(slots), dim3(256), lds, s, f1, f2, out, C, N, B);
    return 0;
}

// schedule parameters (host-computed, kernel argument by value); everything per PAIR unless noted.  Cells are 64 x 64.
struct Sched {
    int Nc, Gb;                 // cells per dimension, big-tile grid per dimension (Nc / 2)
    int n_big_pp, full_rows, rem;   // 128x128 tiles: row-major, `full_rows` full rows of Gb + `rem` in the next row
    int e, A, Bc, U_pp;         // vertical 2-cell units not covered by big tiles: e per full row (A total), Bc in the partial row
    int n_med_pp, n_small_pp;
    int R_b, R_m, R_s;          // rounds of each kind (every workgroup takes one item per round)
    int B;
};

__device__ __forceinline__ void unit_coords(const Sched& S, int u, int& tm, int& c) {
    if (u < S.A) { tm = u / S.e; c = 2 * S.Gb + (u - tm * S.e); return; }
    u -= S.A;
    if (u < S.Bc) { tm = S.full_rows; c = 2 * S.rem + u; return; }
    u -= S.Bc;
    const int r = u / S.Nc;
    tm = S.full_rows + (S.rem > 0 ? 1 : 0) + r;
    c = u - r * S.Nc;
}


// Big (128 x 128) tiles of one workgroup as ONE continuous K stream: the register-staged 2-deep prefetch runs across tile
// boundaries (the last two loads of a tile fetch the first two K steps of the NEXT tile), so a tile switch costs no cold
// prologue: [last MFMA group of tile r] -> LDS store of tile r+1's first K step -> barrier -> epilogue stores of tile r
// (fire and forget, beside the loads already in flight) -> MFMAs of tile r+1.
template <int BK, class NextFn>
__device__ __forceinline__ void big_stream(const float* __restrict__ f1, const float* __restrict__ f2, float* __restrict__ out, int C,
                                           int N, size_t fsz, size_t osz, int n_items, NextFn item /* r -> (b, m0, n0) */, float* smem) {
    constexpr int NP = BK * 32 / 256;                      // float4 per thread per operand per stage
    constexpr int STAGE = BK * 256;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1, kh = lane >> 5, li = lane & 31;
    const int nk = C / BK;
    const int lrow = t >> 5, lcol = (t & 31) * 4;
    const unsigned loff = lrow * N + lcol;
    const int pcol = ((lcol >> 5) & 1) * 64 + (lcol >> 6) * 32 + (lcol & 31);
    float* wa = smem + lrow * 128 + pcol;
    float* wb = smem + BK * 128 + lrow * 128 + pcol;
    const float* fa = smem + kh * 128 + wm * 32 + li;
    const float* fb = smem + BK * 128 + kh * 128 + wn * 32 + li;
    f32x16 acc[2][2];
    auto zero = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    f32x4 ra[2][NP], rb[2][NP];
    auto gload = [&](auto SET, const float* Au, const float* Bu, int k0) {
        constexpr int S = decltype(SET)::value;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            ra[S][p] = *reinterpret_cast<const f32x4*>(Au + (size_t)(k0 + 8 * p) * N + loff);
            rb[S][p] = *reinterpret_cast<const f32x4*>(Bu + (size_t)(k0 + 8 * p) * N + loff);
        }
    };
    auto sstore = [&](auto SET, int buf) {
        constexpr int S = decltype(SET)::value;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            *reinterpret_cast<f32x4*>(wa + buf * STAGE + p * 8 * 128) = ra[S][p];
            *reinterpret_cast<f32x4*>(wb + buf * STAGE + p * 8 * 128) = rb[S][p];
        }
    };
    auto mma = [&](int buf) {
        const float* qa = fa + buf * STAGE;
        const float* qb = fb + buf * STAGE;
        float a[2][2], b[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { a[0][i] = qa[i * 64]; b[0][i] = qb[i * 64]; }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
            if (kk + 2 < BK) {
#pragma unroll
                for (int i = 0; i < 2; ++i) { a[nxt][i] = qa[(kk + 2) * 128 + i * 64]; b[nxt][i] = qb[(kk + 2) * 128 + i * 64]; }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    if (n_items <= 0) return;
    int b, m0, n0;
    item(0, b, m0, n0);
    const float* Au = f1 + b * fsz + m0;
    const float* Bu = f2 + b * fsz + n0;
    float* Ou = out + b * osz + (size_t)m0 * N + n0;
    gload(S0{}, Au, Bu, 0);
    gload(S1{}, Au, Bu, BK);
    __syncthreads();
    sstore(S0{}, 0);
    __syncthreads();
    zero();
    const unsigned so = (wm * 64 + 4 * kh) * N + wn * 64 + li;
#pragma unroll 1
    for (int r = 0; r < n_items; ++r) {
        // the stream continues into the next tile (or re-reads this one's last K steps after the final tile)
        const float *An = Au, *Bn = Bu;
        float* On = Ou;
        const bool more = r + 1 < n_items;
        if (more) {
            item(r + 1, b, m0, n0);
            An = f1 + b * fsz + m0;
            Bn = f2 + b * fsz + n0;
            On = out + b * osz + (size_t)m0 * N + n0;
        }
#pragma unroll 1
        for (int kt = 0; kt < nk; kt += 2) {
            const bool last = kt + 2 >= nk;
            gload(S0{}, last ? An : Au, last ? Bn : Bu, last ? (more ? 0 : C - BK) : (kt + 2) * BK);
            mma(0);
            sstore(S1{}, 1);
            __syncthreads();
            gload(S1{}, last ? An : Au, last ? Bn : Bu, last ? (more ? BK : C - BK) : (kt + 3) * BK);
            mma(1);
            sstore(S0{}, 0);
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                float* p = Ou + (size_t)(i * 32 + (rr & 3) + 8 * (rr >> 2)) * N;
#pragma unroll
                for (int j = 0; j < 2; ++j) __builtin_nontemporal_store(acc[i][j][rr], p + j * 32 + so);
            }
        zero();
        Au = An; Bu = Bn; Ou = On;
    }
}

// MODE bit 0: big tiles as one continuous stream; bit 1: the second workgroup of each CU (slots >= slots/2) starts with its small
// tile, de-phasing the two co-resident workgroups by a quarter tile
template <int MODE>
__global__ __launch_bounds__(256) void gemm_sched2(const float* __restrict__ f1, const float* __restrict__ f2, float* __restrict__ out,
                                                    int C, int N, Sched S) {
    __shared__ __attribute__((aligned(16))) float smem[2 * 16 * 256];
    const int slots = gridDim.x;
    const int s = blockIdx.x;
    const size_t fsz = (size_t)C * N, osz = (size_t)N * N;
    const int lin = (s & 7) * (slots >> 3) + (s >> 3);
    auto smalls = [&](int r0, int r1) {
#pragma unroll 1
        for (int r = r0; r < r1; ++r) {
            const int idx = r * slots + lin;
            if (idx < S.n_small_pp * S.B) {
                const int b = idx / S.n_small_pp, sm = idx - b * S.n_small_pp;
                const int from_units = 2 * (S.U_pp - S.n_med_pp);
                int m0, n0;
                if (sm < from_units) {
                    int tm, c;
                    unit_coords(S, S.n_med_pp + (sm >> 1), tm, c);
                    m0 = tm * 128 + (sm & 1) * 64;
                    n0 = c * 64;
                } else {
                    m0 = (S.Nc - 1) * 64;
                    n0 = (sm - from_units) * 64;
                }
                tile_body<1, 1, false>(f1 + b * fsz, f2 + b * fsz, out + b * osz, C, N, N, m0, n0, smem);
            }
        }
    };
    const bool late = (MODE & 2) && s >= (slots >> 1) && S.R_s > 0;
    if (late) smalls(0, 1);
    if (MODE & 1) {
        big_stream<16>(f1, f2, out, C, N, fsz, osz, S.R_b, [&](int r, int& b, int& m0, int& n0) {
            const int idx = r * slots + lin;
            b = idx / S.n_big_pp;
            const int q = idx - b * S.n_big_pp;
            const int tm = q / S.Gb;
            m0 = tm * 128;
            n0 = (q - tm * S.Gb) * 128;
        }, smem);
    } else {
#pragma unroll 1
        for (int r = 0; r < S.R_b; ++r) {
            const int idx = r * slots + lin;
            const int b = idx / S.n_big_pp, q = idx - b * S.n_big_pp;
            const int tm = q / S.Gb, tn = q - tm * S.Gb;
            tile_body<2, 2, false>(f1 + b * fsz, f2 + b * fsz, out + b * osz, C, N, N, tm * 128, tn * 128, smem);
        }
    }
#pragma unroll 1
    for (int r = 0; r < S.R_m; ++r) {
        const int idx = r * slots + lin;
        if (idx < S.n_med_pp * S.B) {
            const int b = idx / S.n_med_pp, u = idx - b * S.n_med_pp;
            int tm, c;
            unit_coords(S, u, tm, c);
            tile_body<2, 1, false>(f1 + b * fsz, f2 + b * fsz, out + b * osz, C, N, N, tm * 128, c * 64, smem);
        }
    }
    smalls(late ? 1 : 0, S.R_s);
}

template <int DUMMY>
__global__ __launch_bounds__(256) void gemm_sched(const float* __restrict__ f1, const float* __restrict__ f2, float* __restrict__ out,
                                                   int C, int N, Sched S) {
    __shared__ __attribute__((aligned(16))) float smem[2 * 16 * 256];
    const int slots = gridDim.x;
    const int s = blockIdx.x;
    const size_t fsz = (size_t)C * N, osz = (size_t)N * N;
    // inside a round XCD x (= s & 7) owns a contiguous eighth of the items
    const int lin = (s & 7) * (slots >> 3) + (s >> 3);
#pragma unroll 1
    for (int r = 0; r < S.R_b; ++r) {
        const int idx = r * slots + lin;
        const int b = idx / S.n_big_pp, q = idx - b * S.n_big_pp;
        const int tm = q / S.Gb, tn = q - tm * S.Gb;
        tile_body<2, 2, false>(f1 + b * fsz, f2 + b * fsz, out + b * osz, C, N, N, tm * 128, tn * 128, smem);
    }
#pragma unroll 1
    for (int r = 0; r < S.R_m; ++r) {
        const int idx = r * slots + lin;
        if (idx < S.n_med_pp * S.B) {
            const int b = idx / S.n_med_pp, u = idx - b * S.n_med_pp;
            int tm, c;
            unit_coords(S, u, tm, c);
            tile_body<2, 1, false>(f1 + b * fsz, f2 + b * fsz, out + b * osz, C, N, N, tm * 128, c * 64, smem);
        }
    }
#pragma unroll 1
    for (int r = 0; r < S.R_s; ++r) {
        const int idx = r * slots + lin;
        if (idx < S.n_small_pp * S.B) {
            const int b = idx / S.n_small_pp, sm = idx - b * S.n_small_pp;
            const int from_units = 2 * (S.U_pp - S.n_med_pp);
            int m0, n0;
            if (sm < from_units) {
                int tm, c;
                unit_coords(S, S.n_med_pp + (sm >> 1), tm, c);
                m0 = tm * 128 + (sm & 1) * 64;
                n0 = c * 64;
            } else {
                m0 = (S.Nc - 1) * 64;
                n0 = (sm - from_units) * 64;
            }
            tile_body<1, 1, false>(f1 + b * fsz, f2 + b * fsz, out + b * osz, C, N, N, m0, n0, smem);
        }
    }
}

static bool make_sched(int N, int B, int slots, Sched& S) {
    if (N % 64 || B < 1) return false;
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
    // every cell covered exactly once?
    return 4L * S.n_big_pp + 2L * S.n_med_pp + S.n_small_pp == (long)S.Nc * S.Nc;
}

}  // namespace

static unsigned* g_extq = nullptr;
extern "C" void gemm3_set_queue(unsigned* q) { g_extq = q; }
extern "C" int gemm3_launch(const float* f1, const float* f2, float* out, int B, int C, int N, int mode, hipStream_t s) {
    const int tiles = (N + 127) / 128;
    dim3 grid(tiles * tiles, 1, B), block(256);
    switch (mode) {
        case 0: return mv_corr_volume(f1, f2, out, B, C, N, N, MV_F32, MV_LAYOUT_CHW, s);
        case 1: hipLaunchKernelGGL(gemm_perm<1>, grid, block, 0, s, f1, f2, out, C, N, N, tiles, tiles); break;
        case 2: hipLaunchKernelGGL(gemm_perm<4>, grid, block, 0, s, f1, f2, out, C, N, N, tiles, tiles); break;
        case 10: return launch_uni<2, 2, 1>(f1, f2, out, B, C, N, 1024, s);
        case 11: return launch_uni<2, 2, 1>(f1, f2, out, B, C, N, 768, s);
        case 12: return launch_uni<2, 2, 1>(f1, f2, out, B, C, N, 512, s);
        case 13: return launch_uni<2, 2, 1>(f1, f2, out, B, C, N, 256, s);
        case 20: return launch_uni<4, 2, 1>(f1, f2, out, B, C, N, 512, s);
        case 21: return launch_uni<4, 2, 1>(f1, f2, out, B, C, N, 256, s);
        case 22: return launch_uni<2, 4, 1>(f1, f2, out, B, C, N, 512, s);
        case 23: return launch_uni<2, 4, 1>(f1, f2, out, B, C, N, 256, s);
        case 30: return launch_uni<4, 4, 1>(f1, f2, out, B, C, N, 256, s);
        case 50: return launch_uni<2, 2, 1, 32>(f1, f2, out, B, C, N, 512, s);
        case 51: return launch_uni<2, 2, 1, 32>(f1, f2, out, B, C, N, 256, s);
        case 52: return launch_uni<2, 2, 1, 64>(f1, f2, out, B, C, N, 256, s);
        case 53: return launch_uni<2, 2, 1, 8>(f1, f2, out, B, C, N, 512, s);
        case 60: return launch_uni<2, 2, 1, 16, 1>(f1, f2, out, B, C, N, 512, s);
        case 61: return launch_uni<2, 2, 1, 16, 2>(f1, f2, out, B, C, N, 512, s);
        case 62: return launch_uni<2, 2, 1, 16, 3>(f1, f2, out, B, C, N, 512, s);
        case 63: return launch_uni<2, 2, 1, 32, 3>(f1, f2, out, B, C, N, 512, s);
        case 64: return launch_uni<2, 2, 1, 16, 3>(f1, f2, out, B, C, N, 768, s);
        case 65: return launch_uni<2, 2, 1, 16, 3>(f1, f2, out, B, C, N, 256, s);
        case 66: return launch_uni<2, 2, 1, 32, 3>(f1, f2, out, B, C, N, 256, s);
        case 40: case 41: case 42: case 43: case 44: case 45: case 46: case 47: {
            Sched S;
            const int slots = (mode & 4) ? 768 : 512;
            if (!make_sched(N, B, slots, S)) return -2;
            switch (mode & 3) {
                case 0: hipLaunchKernelGGL(gemm_sched2<0>, dim3(slots), block, 0, s, f1, f2, out, C, N, S); break;
                case 1: hipLaunchKernelGGL(gemm_sched2<1>, dim3(slots), block, 0, s, f1, f2, out, C, N, S); break;
                case 2: hipLaunchKernelGGL(gemm_sched2<2>, dim3(slots), block, 0, s, f1, f2, out, C, N, S); break;
                default: hipLaunchKernelGGL(gemm_sched2<3>, dim3(slots), block, 0, s, f1, f2, out, C, N, S); break;
            }
            break;
        }
        case 3: case 4: case 5: case 6: case 7: {
            Sched S;
            const int slots = mode == 3 ? 1024 : mode == 4 ? 768 : mode == 5 ? 512 : mode == 6 ? 384 : 256;
            if (!make_sched(N, B, slots, S)) return -2;
            hipLaunchKernelGGL(gemm_sched<0>, dim3(slots), block, 0, s, f1, f2, out, C, N, S);
            break;
        }
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
