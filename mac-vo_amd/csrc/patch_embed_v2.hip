// (f)2 — the fused cost patch embedding for ANY slice size (round 5): strip-mined.
//
//   cost_maps [S, 1, H2, W2] (fp32 or the 16-bit operand type) -> F.pad to multiples of 8 -> Conv2d(1, 16, 6, 2, 2) -> ReLU -> Conv2d(16, 32, 6, 2, 2) -> ReLU
//   -> Conv2d(32, 64, 6, 2, 2) -> [S, 64, H2/8, W2/8] or token-major [S, H2/8 * W2/8, 64]   (FlowFormer PatchEmbed.proj, patch_size 8; see patch_embed.hip for
//   the reference citations: flownet.py:26, covhead.py:61-64, Config/Train/Demo.yaml:20-36; source absent from the checkout, parity pinned to F.conv2d).
//
// patch_embed.hip keeps a whole 60 / 64 x 80 slice and both intermediate maps in LDS (two slices per pass).  That plan does not exist for the other
// sizes the reference runs: 90 x 160 slices (1280 x 720 frames, BASELINE configs[2]: the conv1 map alone is 140 KB) and 80 x 80 (the 640 x 640 fixture,
// DataLoader/Dataset/TartanAir2.py:82-85: 100 tokens = 6.25 tiles, 25 conv2 tiles).  Here a workgroup walks a slice in STRIPS of R3 token rows:
//
//   strip = token rows [y3a, y3a + R3)  <-  conv2 rows [2 y3a - 2, 2 y3a + 2 R3 + 2)  <-  conv1 rows 2 x (...) + 4  <-  input rows 2 x (...) + 4 (+ 2)
//
//   every layer's LDS buffer is a WINDOW of the zero-padded map (window row = map row - r0); only the rows of the window that exist in the map are
//   computed, the others are zeroed, so the halo rows between strips are recomputed (90 x 160, R3 = 4: conv2 x 1.33, conv1 x 1.5: + 17 % of the FLOPs) and a
//   slice that fits (80 x 80, 60 x 80: one strip) pays nothing.  No state is carried between strips: (slice, strip) items are spread over the persistent
//   workgroups like slices were.
//
// Implicit GEMMs as in patch_embed.hip (v_mfma_f32_16x16x32, K = taps x input channels, cin innermost, maps stored [8-channel chunk][column parity][row]
// [column / 2][16 B]) with three changes that the LDS-cycle model of that kernel asked for (profiles/probes/r5_pe_lds_sim.py: the conv1 / conv2 epilogues'
// 2-byte stores and conv1's dword reads were 2-way conflicted and half of all LDS cycles):
//   * conv1 / conv2 run with the operands SWAPPED (mma16t): a lane ends with four consecutive CHANNELS of one pixel = one 8-byte store into the pixel's
//     cell instead of four 2-byte stores into four cells;
//   * conv1 walks whole row groups (G rows with G W1 a multiple of 16: five tiles) whose per-lane fragment / store offsets are computed once per kernel;
//   * conv3 splits K (the 36 taps) over wave pairs instead of pairing two slices: each activation fragment is read once per N pair, the halves meet in LDS.
#include "patch_embed_dev.h"
#include <algorithm>
#include <atomic>

using namespace pe;

namespace {

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int cmin(int a, int b) { return a < b ? a : b; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int H2_, int W2_, int R3_>
struct PG {
    static constexpr int H2 = H2_, W2 = W2_, R3 = R3_;
    static constexpr int HP = cdiv(H2, 8) * 8, WP = cdiv(W2, 8) * 8;
    static constexpr int H1 = HP / 2, W1 = WP / 2, Hc = HP / 4, Wc = WP / 4, H3 = HP / 8, W3 = WP / 8, M3 = H3 * W3;
    static constexpr int NS = cdiv(H3, R3);                                    // strips per slice
    // windows (rows): conv2 map, conv1 map, input
    static constexpr int ROWS2 = 2 * R3 + 4;
    static constexpr int MAXC2 = cmin(Hc, ROWS2);                               // real conv2 rows of a strip, at most
    static constexpr int G1 = (W1 % 16 == 0) ? 1 : (W1 % 8 == 0) ? 2 : 4;       // conv1 row group: G1 W1 = a whole number of 16-pixel tiles
    static_assert((G1 * W1) % 16 == 0, "W2 must be a multiple of 8");
    static constexpr int TG1 = G1 * W1 / 16;                                    // tiles per row group
    static constexpr int MAXC1 = cmin(H1, 2 * MAXC2 + 4);                       // real conv1 rows of a strip, at most
    static constexpr int ROWS1 = 2 * MAXC2 + 4;
    static constexpr int ROWSIN = 2 * cdiv(MAXC1, G1) * G1 + 6;                 // + 4 halo, + 2 for the zero-weight padding taps (ky = 6, 7)
    // elements; halo 2 + the zero-weight taps (kx' = 6, 7); pitch / 2 = 4 (mod 8) dwords: the two input rows of a ds_read_b32 service group (four rows apart,
    // see mv_patch_embed_pack) sit 16 banks apart
    static constexpr int in_pitch(int lo) { int x = lo; while ((x / 2) % 8 != 4) x += 2; return x; }
    static constexpr int IN_PITCH = in_pitch(WP + 8);
    static constexpr int O1_COLS = W1 + 4, O2_COLS = Wc + 4;
    static_assert(O1_COLS % 2 == 0 && O2_COLS % 2 == 0, "column-parity planes");
    // row pitch of a parity plane in 16-byte cells: the smallest pitch >= the row with pitch % 8 == r, r from the LDS-cycle model
    // (profiles/probes/r5_pe_v2_index_model.py)
    static constexpr int pad_xh(int lo, int r) { int x = lo; while (x % 8 != r) ++x; return x; }
    // a 16-lane read group walks up to two consumer rows of L pixels (L = Wc for the conv1 map, W3 for the conv2 map), 2 x pitch cells apart: the groups tile
    // the 16 slots of a bank row when 2 x pitch = L (mod 16): L = 20 -> pitch = 2 (mod 8), 10 -> 5, 40 -> 4 (model: 1.00x for all three geometries)
    static_assert(Wc % 2 == 0 && W3 % 2 == 0, "even consumer rows");
    static constexpr int O1_XH = pad_xh(O1_COLS / 2, (Wc % 16) / 2);
    static constexpr int O2_XH = pad_xh(O2_COLS / 2, (W3 % 16) / 2);
    static constexpr unsigned plane_pad(unsigned b) { return b + ((128u + 256u - b % 256u) % 256u); }   // -> = 128 (mod 256): the four 16-lane groups of a
    static constexpr unsigned O1_PLANE = plane_pad(ROWS1 * O1_XH * 16), O2_PLANE = plane_pad(ROWS2 * O2_XH * 16);   // b128 read sit in different planes
    static constexpr unsigned o1_cell(int c, int row, int col) { return (unsigned)(c * 2 + (col & 1)) * O1_PLANE + (unsigned)(row * O1_XH + (col >> 1)) * 16; }
    static constexpr unsigned o2_cell(int c, int row, int col) { return (unsigned)(c * 2 + (col & 1)) * O2_PLANE + (unsigned)(row * O2_XH + (col >> 1)) * 16; }
    static constexpr unsigned OFF_IN0 = 0, IN0_BYTES = (ROWSIN * IN_PITCH * 2 + 255) / 256 * 256;
    static constexpr unsigned OFF_O1 = OFF_IN0 + IN0_BYTES, O1_BYTES = 4 * O1_PLANE;
    static constexpr unsigned OFF_O2 = OFF_O1 + O1_BYTES, O2_BYTES = 8 * O2_PLANE;
    static constexpr unsigned OFF_W2B = OFF_O2 + O2_BYTES, W2B_BYTES = 18 * 1024;
    static constexpr unsigned LDS_BYTES = OFF_W2B + W2B_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS plan");
    // tile counts (16 pixels / tokens each)
    static constexpr int NT2 = cdiv(MAXC2 * Wc, 16), NT2W = cdiv(NT2, 4);       // conv2: tiles of a strip / per wave (round robin)
    static constexpr int NT3 = cdiv(cmin(R3, H3) * W3, 16);                     // conv3: every wave walks all of them (K split, N pair)
    // conv3's K halves meet in the dead input + conv1 region: [N pair 2][tile][N tile 2][64 lanes] f32x4
    static constexpr unsigned RED_BYTES = 2u * NT3 * 2 * 64 * 16;
    static_assert(RED_BYTES <= IN0_BYTES + O1_BYTES, "K-half exchange area");
    static constexpr int EPL32 = 4, EPL16 = 8;
    static_assert(W2 % 8 == 0, "16-byte staging loads of 16-bit cells");
};

// One (slice, strip) item.  All row quantities are uniform (SGPRs).
struct Strip {
    int slice, y3a, r3s;          // token rows [y3a, y3a + r3s)
    int r0_2, c2lo, c2n;          // conv2 window origin; real rows [c2lo, c2lo + c2n)
    int r0_1, c1lo, c1n;          // conv1 window origin; real rows
    int r0_in, nin;               // input window origin; rows to stage
};

template <typename P>
__device__ __forceinline__ Strip make_strip(int item) {
    Strip s;
    s.slice = item / P::NS;
    const int k = item - s.slice * P::NS;
    s.y3a = k * P::R3;
    s.r3s = min(P::R3, P::H3 - s.y3a);
    s.r0_2 = 2 * s.y3a - 2;
    s.c2lo = max(0, s.r0_2);
    s.c2n = min(P::Hc, 2 * s.y3a + 2 * s.r3s + 2) - s.c2lo;
    s.r0_1 = 2 * s.c2lo - 2;
    s.c1lo = max(0, s.r0_1);
    s.c1n = min(P::H1, 2 * (s.c2lo + s.c2n) + 2) - s.c1lo;
    s.r0_in = 2 * s.c1lo - 2;
    s.nin = 2 * (cdiv(s.c1n, P::G1) * P::G1) + 6;
    return s;
}

template <int H2, int W2, int R3, bool TOKENS, bool F16, bool IN16, bool OUT16>
__global__ __launch_bounds__(256) void cost_patch_embed_strip_kernel(const void* __restrict__ vol_, const char* __restrict__ wp, void* __restrict__ out_, int S) {
    using P = PG<H2, W2, R3>;
    extern __shared__ __attribute__((aligned(16))) char smem_pe2[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n16 = lane & 15, g4 = lane >> 4;
    char* const in0 = smem_pe2 + P::OFF_IN0;
    char* const o1 = smem_pe2 + P::OFF_O1;
    char* const o2 = smem_pe2 + P::OFF_O2;
    const char* const vol = reinterpret_cast<const char*>(vol_);
    constexpr int EPL = IN16 ? P::EPL16 : P::EPL32;                           // cells per 16-byte load
    constexpr int RQ = W2 / EPL;                                              // loads per input row
    constexpr int NPRE = cdiv(P::ROWSIN * RQ, 256);
    constexpr int ESZ = IN16 ? 2 : 4;

    // ---- once per workgroup: zero everything (column halos and the padding columns stay zero for good), weights + biases -> registers / LDS
    for (unsigned a = (unsigned)t * 16u; a < P::OFF_W2B; a += 256u * 16u) *reinterpret_cast<i32x4*>(smem_pe2 + a) = i32x4{0, 0, 0, 0};
    i32x4 w2f[18];                                                            // conv2, channel tile 0: registers; tile 1: LDS (as patch_embed.hip)
#pragma unroll
    for (int ks = 0; ks < 18; ++ks) w2f[ks] = *reinterpret_cast<const i32x4*>(wp + PE_W2_OFF + ((size_t)ks * 64 + lane) * 16);
    for (unsigned a = (unsigned)t * 16u; a < P::W2B_BYTES; a += 256u * 16u)
        *reinterpret_cast<i32x4*>(smem_pe2 + P::OFF_W2B + a) = *reinterpret_cast<const i32x4*>(wp + PE_W2_OFF + 18 * 1024 + a);
    i32x4 w1f[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) w1f[s] = *reinterpret_cast<const i32x4*>(wp + PE_W1_OFF + (s * 64 + lane) * 16);
    const float* bias = reinterpret_cast<const float*>(wp + PE_B_OFF);
    float b1v[4], b2v[2][4];                                                  // swapped products: a lane owns channels 4 g4 + e of its tile
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        b1v[e] = bias[4 * g4 + e];
        b2v[0][e] = bias[32 + 4 * g4 + e];
        b2v[1][e] = bias[48 + 4 * g4 + e];
    }
    const int kh = wave & 1, np = wave >> 1;                                  // conv3: K half (taps 18 kh ..), pair of 16-channel tiles
    float b3s[2][4], b3n[2];                                                  // token-major (swapped) / channel-major (plain) epilogue
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        b3n[j] = bias[64 + (2 * np + j) * 16 + n16];
#pragma unroll
        for (int e = 0; e < 4; ++e) b3s[j][e] = bias[64 + (2 * np + j) * 16 + 4 * g4 + e];
    }
    const char* const w3 = wp + PE_W3_OFF + ((size_t)(2 * np) * 36 * 64 + lane) * 16;   // (j, tap) at + (j * 36 + tap) KB

    // conv1: per-lane offsets of the TG1 tiles of a row group (fragment in the input window / 8-byte store in the conv1 window), once per kernel
    int c1_a[P::TG1], c1_d[P::TG1], c1_rr[P::TG1];
#pragma unroll
    for (int j = 0; j < P::TG1; ++j) {
        const int pix = j * 16 + n16, rr = pix / P::W1, xx = pix - rr * P::W1;
        c1_rr[j] = rr;
        c1_a[j] = ((2 * rr + 4 * (g4 & 1) + (g4 >> 1)) * P::IN_PITCH + 2 * xx) * 2;   // ky of k-step 0: 4 (g & 1) + (g >> 1); k-step 1: + 2 (mv_patch_embed_pack)
        c1_d[j] = (int)P::o1_cell(g4 >> 1, rr, xx + 2) + (g4 & 1) * 8;
    }
    __syncthreads();

    // staging: this thread's 16-byte pieces of the NEXT item's input window travel in registers while the current one is convolved
    i32x4 pre[NPRE];
    auto fetch = [&](int item) {
        const bool live = item < S * P::NS;
        const Strip sn = make_strip<P>(live ? item : 0);
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int q = t + 256 * i;
            const int row = q / RQ, y = sn.r0_in + row;
            const bool ok = live && row < sn.nin && y >= 0 && y < H2;
            pre[i] = ok ? __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(vol + ((size_t)sn.slice * (H2 * W2) + (size_t)y * W2) * ESZ) + (q - row * RQ))
                        : i32x4{0, 0, 0, 0};
        }
    };

    const int n_items = S * P::NS;
    int item = blockIdx.x;
    if (item < n_items) fetch(item);
    for (; item < n_items; item += gridDim.x) {
        const Strip st = make_strip<P>(item);
        int gv = g4;                                      // opaque copy: keeps hipcc from hoisting the epilogues' address arithmetic out of the item loop
        asm volatile("" : "+v"(gv));
        // ---- (A) rows of the two map windows that are outside the map for this strip -> zero (the previous item left data there); input window <- registers
        for (int r = 0; r < P::ROWS1; ++r) {
            const int y1 = st.r0_1 + r;
            if (y1 >= st.c1lo && y1 < st.c1lo + st.c1n) continue;            // (uniform)
            for (int c = t; c < 4 * P::O1_XH; c += 256) {
                const int pl = c / P::O1_XH, x = c - pl * P::O1_XH;
                *reinterpret_cast<i32x4*>(o1 + pl * P::O1_PLANE + (r * P::O1_XH + x) * 16) = i32x4{0, 0, 0, 0};
            }
        }
        for (int r = 0; r < P::ROWS2; ++r) {
            const int y2 = st.r0_2 + r;
            if (y2 >= st.c2lo && y2 < st.c2lo + st.c2n) continue;
            for (int c = t; c < 8 * P::O2_XH; c += 256) {
                const int pl = c / P::O2_XH, x = c - pl * P::O2_XH;
                *reinterpret_cast<i32x4*>(o2 + pl * P::O2_PLANE + (r * P::O2_XH + x) * 16) = i32x4{0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int q = t + 256 * i;
            const int row = q / RQ, x = EPL * (q - row * RQ);
            if (row < st.nin) {
                unsigned* d = reinterpret_cast<unsigned*>(in0 + (row * P::IN_PITCH + x + 2) * 2);
                if constexpr (IN16) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e] = (unsigned)pre[i][e];
                } else {
                    const f32x4 v = __builtin_bit_cast(f32x4, pre[i]);
                    d[0] = cvt_pack<F16>(v[0], v[1]);
                    d[1] = cvt_pack<F16>(v[2], v[3]);
                }
            }
        }
        fetch(item + (int)gridDim.x);
        __syncthreads();

        // ---- (B) conv1: row groups round robin over the waves; a group = G1 rows = TG1 tiles; K = 2 x 32 = (ky 0..7) x (kx' 0..7), taps >= 6 carry zero weights
        {
            // software-pipelined by hand (see patch_embed_v3.hip): the fragment of the NEXT tile is read before this tile's store — hipcc keeps LDS reads
            // behind earlier LDS writes it cannot disambiguate, which serialised the tiles into one LDS round trip each
            const int ngroups = (st.c1n + P::G1 - 1) / P::G1;
            auto load_tile = [&](int g, int j, i32x4* af) __attribute__((always_inline)) {
                const char* a0 = in0 + (2 * g * P::G1) * (P::IN_PITCH * 2) + c1_a[j];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const unsigned* ap = reinterpret_cast<const unsigned*>(a0 + 2 * s * P::IN_PITCH * 2);
                    af[s] = i32x4{(int)ap[0], (int)ap[1], (int)ap[2], (int)ap[3]};
                }
            };
            i32x4 afr[2][2];
            if (wave < ngroups) load_tile(wave, 0, afr[0]);
            for (int g = wave; g < ngroups; g += 4) {
                char* dbase = o1 + ((st.c1lo - st.r0_1 + g * P::G1) * P::O1_XH) * 16;
#pragma unroll
                for (int j = 0; j < P::TG1; ++j) {
                    // next tile: j + 1 of this group, or tile 0 of this wave's next group
                    i32x4* const mine = afr[j & 1];
                    i32x4* const next = afr[(j + 1) & 1];
                    if (j + 1 < P::TG1) load_tile(g, j + 1, next);
                    else if (g + 4 < ngroups) load_tile(g + 4, 0, next);
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 2; ++s) acc = mma16t<F16>(__builtin_bit_cast(bf16x8, mine[s]), __builtin_bit_cast(bf16x8, w1f[s]), acc);
                    if (P::G1 == 1 || g * P::G1 + c1_rr[j] < st.c1n) {
                        unsigned* d = reinterpret_cast<unsigned*>(dbase + c1_d[j]);
                        const unsigned lo = cvt_pack<F16>(fmaxf(acc[0] + b1v[0], 0.f), fmaxf(acc[1] + b1v[1], 0.f));
                        const unsigned hi = cvt_pack<F16>(fmaxf(acc[2] + b1v[2], 0.f), fmaxf(acc[3] + b1v[3], 0.f));
                        *reinterpret_cast<unsigned long long*>(d) = (unsigned long long)lo | ((unsigned long long)hi << 32);
                    }
                }
                if (P::TG1 & 1) {   // an odd tile count flips the ring parity for the next group: move the prefetched fragment to slot 0
                    afr[0][0] = afr[1][0];
                    afr[0][1] = afr[1][1];
                }
            }
        }
        __syncthreads();

        // ---- (C) conv2: tiles round robin (slot i of this wave = tile wave + 4 i); a k-step = two neighbouring taps (same ky, kx = 2 (ks % 3) + h) x 16 input
        // channels: lane group l / 16 = (h, channel chunk) reads chunk (l / 16) % 2 of the parity-h plane
        {
            const int n2 = st.c2n * P::Wc;
            f32x4 acc[P::NT2W][2];
            const char* abase[P::NT2W];
#pragma unroll
            for (int i = 0; i < P::NT2W; ++i) {
                acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int p = min((wave + 4 * i) * 16 + n16, n2 - 1), oy = p / P::Wc, ox = p - oy * P::Wc;
                abase[i] = o1 + P::o1_cell(g4 & 1, 2 * oy, 2 * ox) + (g4 >> 1) * P::O1_PLANE;
            }
            constexpr int PFA = 1;                                      // fragments are fetched PFA k-steps ahead of their MFMAs
            bf16x8 af[PFA + 1][P::NT2W], bf1[PFA + 1];
            const char* const wb1 = smem_pe2 + P::OFF_W2B + lane * 16;
            auto fetch_k = [&](int ks) __attribute__((always_inline)) {
                bf1[ks % (PFA + 1)] = *reinterpret_cast<const bf16x8*>(wb1 + ks * 1024);
#pragma unroll
                for (int i = 0; i < P::NT2W; ++i)
                    af[ks % (PFA + 1)][i] = *reinterpret_cast<const bf16x8*>(abase[i] + ((ks / 3) * P::O1_XH + ks % 3) * 16);
            };
#pragma unroll
            for (int ks = 0; ks < PFA; ++ks) fetch_k(ks);
#pragma unroll
            for (int ks = 0; ks < 18; ++ks) {
                if (ks + PFA < 18) fetch_k(ks + PFA);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < P::NT2W; ++i) {
                    acc[i][0] = mma16t<F16>(af[ks % (PFA + 1)][i], __builtin_bit_cast(bf16x8, w2f[ks]), acc[i][0]);
                    acc[i][1] = mma16t<F16>(af[ks % (PFA + 1)][i], bf1[ks % (PFA + 1)], acc[i][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // epilogue: a lane holds channels nt * 16 + 4 g4 + e of pixel n16 of the tile: one 8-byte store per channel tile
            const int rowoff = st.c2lo - st.r0_2;
#pragma unroll
            for (int i = 0; i < P::NT2W; ++i) {
                const int pp = (wave + 4 * i) * 16 + n16;
                if (pp < n2) {
                    const int y = pp / P::Wc, x = pp - y * P::Wc;
                    char* d = o2 + (gv >> 1) * 2 * P::O2_PLANE + (gv & 1) * 8 + P::o2_cell(0, y + rowoff, x + 2);
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const unsigned lo = cvt_pack<F16>(fmaxf(acc[i][nt][0] + b2v[nt][0], 0.f), fmaxf(acc[i][nt][1] + b2v[nt][1], 0.f));
                        const unsigned hi = cvt_pack<F16>(fmaxf(acc[i][nt][2] + b2v[nt][2], 0.f), fmaxf(acc[i][nt][3] + b2v[nt][3], 0.f));
                        *reinterpret_cast<unsigned long long*>(d + nt * 4 * P::O2_PLANE) = (unsigned long long)lo | ((unsigned long long)hi << 32);
                    }
                }
            }
        }
        __syncthreads();

        // ---- (D) conv3 over the strip's tokens: this wave = taps 18 kh .. 18 kh + 17 x channel tiles 2 np, 2 np + 1; a k-step = one tap x 32 input channels
        {
            const int n3 = st.r3s * P::W3;
            f32x4 acc[P::NT3][2];
            const char* abase[P::NT3];
#pragma unroll
            for (int i = 0; i < P::NT3; ++i) {
                acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int q = min(i * 16 + n16, n3 - 1), oy = q / P::W3, ox = q - oy * P::W3;
                abase[i] = o2 + P::o2_cell(g4, 2 * oy, 2 * ox);
            }
            constexpr int PF = 6, PFA = 1;                              // weight fragment pairs in flight (L2 latency); activation fragments ahead
            const int tap0 = 18 * kh;
            i32x4 bq[PF][2];
#pragma unroll
            for (int kk = 0; kk < PF; ++kk)
#pragma unroll
                for (int j = 0; j < 2; ++j) bq[kk][j] = *reinterpret_cast<const i32x4*>(w3 + (size_t)(j * 36 + tap0 + kk) * 1024);
            bf16x8 af[PFA + 1][P::NT3];
            // taps 18 kh + kk: ky = 3 kh + kk / 6, kx = kk % 6
            const unsigned khoff = P::o2_cell(0, 3, 0) * (unsigned)kh;
            auto fetch_a = [&](int kk) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < P::NT3; ++i)
                    af[kk % (PFA + 1)][i] = *reinterpret_cast<const bf16x8*>(abase[i] + khoff + P::o2_cell(0, kk / 6, kk % 6));
            };
#pragma unroll
            for (int kk = 0; kk < PFA; ++kk) fetch_a(kk);
#pragma unroll
            for (int kk = 0; kk < 18; ++kk) {
                const i32x4 b0 = bq[kk % PF][0], b1 = bq[kk % PF][1];
                if (kk + PFA < 18) fetch_a(kk + PFA);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < P::NT3; ++i) {
                    if constexpr (TOKENS) {
                        acc[i][0] = mma16t<F16>(af[kk % (PFA + 1)][i], __builtin_bit_cast(bf16x8, b0), acc[i][0]);
                        acc[i][1] = mma16t<F16>(af[kk % (PFA + 1)][i], __builtin_bit_cast(bf16x8, b1), acc[i][1]);
                    } else {
                        acc[i][0] = mma16<F16>(af[kk % (PFA + 1)][i], __builtin_bit_cast(bf16x8, b0), acc[i][0]);
                        acc[i][1] = mma16<F16>(af[kk % (PFA + 1)][i], __builtin_bit_cast(bf16x8, b1), acc[i][1]);
                    }
                }
                if (kk + PF < 18) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) bq[kk % PF][j] = *reinterpret_cast<const i32x4*>(w3 + (size_t)(j * 36 + tap0 + kk + PF) * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // K halves: the upper half's partial sums cross through LDS (input + conv1 windows are dead), the lower half adds, finishes and stores
            // (and leaves the area zeroed: see below)
            f32x4* const red = reinterpret_cast<f32x4*>(smem_pe2) + (size_t)np * (P::NT3 * 2 * 64) + lane;
            if (kh == 1) {
#pragma unroll
                for (int i = 0; i < P::NT3; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) red[(i * 2 + j) * 64] = acc[i][j];
            }
            __syncthreads();
            if (kh == 0) {
                const size_t tok0 = (size_t)st.slice * P::M3 + (size_t)st.y3a * P::W3;            // first token of the strip (token-major index)
#pragma unroll
                for (int i = 0; i < P::NT3; ++i) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x4 o = red[(i * 2 + j) * 64];
                        red[(i * 2 + j) * 64] = f32x4{0.f, 0.f, 0.f, 0.f};   // the exchange area lies over the column halos of the input / conv1 windows, which every
                        f32x4 v;                                             // item relies on being zero: each entry is read by exactly one lane, which restores it
                        if constexpr (TOKENS) {
                            // lane: token i * 16 + n16, channels (2 np + j) * 16 + 4 g4 + e
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = (acc[i][j][e] + o[e]) + b3s[j][e];
                            const int q = i * 16 + n16;
                            if (q < n3) {
                                const size_t at = (tok0 + q) * 64 + (2 * np + j) * 16 + 4 * gv;
                                if constexpr (OUT16) {
                                    unsigned* d = reinterpret_cast<unsigned*>(reinterpret_cast<uint16_t*>(out_) + at);
                                    d[0] = cvt_pack<F16>(v[0], v[1]);
                                    d[1] = cvt_pack<F16>(v[2], v[3]);
                                } else {
                                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out_) + at) = v;
                                }
                            }
                        } else {
                            // lane: channel (2 np + j) * 16 + n16, tokens i * 16 + 4 g4 + e
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = (acc[i][j][e] + o[e]) + b3n[j];
                            const int q = i * 16 + 4 * gv;
                            const size_t at = ((size_t)st.slice * 64 + (2 * np + j) * 16 + n16) * P::M3 + (size_t)st.y3a * P::W3 + q;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if (q + e < n3) {
                                    if constexpr (OUT16) reinterpret_cast<uint16_t*>(out_)[at + e] = cvt_bits<F16>(v[e]);
                                    else reinterpret_cast<float*>(out_)[at + e] = v[e];
                                }
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();                                              // the exchange area is the next item's input window
    }
}

template <int H2, int W2, int R3, bool F16, bool IN16, bool OUT16>
int launch_strip(const void* cost_maps, const void* packed, void* out, int S, int token_layout, hipStream_t stream) {
    using P = PG<H2, W2, R3>;
    static std::atomic<bool> attr_done[64];
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[dev].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute((const void*)cost_patch_embed_strip_kernel<H2, W2, R3, true, F16, IN16, OUT16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)cost_patch_embed_strip_kernel<H2, W2, R3, false, F16, IN16, OUT16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done[dev].store(true, std::memory_order_release);
    }
    int ncu = cus[dev].load(std::memory_order_relaxed);
    if (!ncu) {
        hipDeviceProp_t prop;
        ncu = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        cus[dev].store(ncu, std::memory_order_relaxed);
    }
    const long items = (long)S * P::NS;
    const dim3 grid((unsigned)std::min<long>(items, ncu));                  // persistent: one workgroup per CU
    if (token_layout)
        hipLaunchKernelGGL((cost_patch_embed_strip_kernel<H2, W2, R3, true, F16, IN16, OUT16>), grid, dim3(256), P::LDS_BYTES, stream, cost_maps, (const char*)packed, out, S);
    else
        hipLaunchKernelGGL((cost_patch_embed_strip_kernel<H2, W2, R3, false, F16, IN16, OUT16>), grid, dim3(256), P::LDS_BYTES, stream, cost_maps, (const char*)packed, out, S);
    return mv_launch_status();
}

template <int H2, int W2, int R3>
int dispatch_strip(const void* cost_maps, int in16, const void* packed, void* out, int out16, int S, int token_layout, int f16, hipStream_t st) {
    if (f16) {
        if (!in16) return launch_strip<H2, W2, R3, true, false, false>(cost_maps, packed, out, S, token_layout, st);
        return out16 ? launch_strip<H2, W2, R3, true, true, true>(cost_maps, packed, out, S, token_layout, st)
                     : launch_strip<H2, W2, R3, true, true, false>(cost_maps, packed, out, S, token_layout, st);
    }
    if (!in16) return launch_strip<H2, W2, R3, false, false, false>(cost_maps, packed, out, S, token_layout, st);
    return out16 ? launch_strip<H2, W2, R3, false, true, true>(cost_maps, packed, out, S, token_layout, st)
                 : launch_strip<H2, W2, R3, false, true, false>(cost_maps, packed, out, S, token_layout, st);
}

}  // namespace

// library-internal (patch_embed.hip dispatches here): MV_ERR_UNSUPPORTED for a size without an instantiation
int mv_cost_patch_embed_strip(const void* cost_maps, int in16, const void* packed, void* out, int out16, int S, int H2, int W2, int token_layout, int f16,
                              mvStream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    if (H2 == 80 && W2 == 80) return dispatch_strip<80, 80, 10>(cost_maps, in16, packed, out, out16, S, token_layout, f16, st);     // one strip: 100 tokens
    if (H2 == 90 && W2 == 160) return dispatch_strip<90, 160, 4>(cost_maps, in16, packed, out, out16, S, token_layout, f16, st);    // 1280x720: 3 strips of 80 tokens
    if (H2 == 96 && W2 == 160) return dispatch_strip<96, 160, 4>(cost_maps, in16, packed, out, out16, S, token_layout, f16, st);    // the padded slice PatchEmbed.forward hands over
    if (H2 == 60 && W2 == 80) return dispatch_strip<60, 80, 8>(cost_maps, in16, packed, out, out16, S, token_layout, f16, st);      // (A/B against patch_embed.hip: MV_PE_STRIP=1)
    if (H2 == 64 && W2 == 80) return dispatch_strip<64, 80, 8>(cost_maps, in16, packed, out, out16, S, token_layout, f16, st);
    return MV_ERR_UNSUPPORTED;
}

int mv_cost_patch_embed_strip_supported(int H2, int W2) {
    return (H2 == 80 && W2 == 80) || ((H2 == 90 || H2 == 96) && W2 == 160) || ((H2 == 60 || H2 == 64) && W2 == 80);
}
