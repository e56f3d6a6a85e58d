// (f)2 — the consumer side of the cost volume: FlowFormer's cost PATCH EMBEDDING, fused  (round 4)
//
//   cost_maps [S = B·H1·W1, 1, H2, W2] fp32 (one H2 x W2 slice per query pixel, what mv_corr_volume writes)
//     -> F.pad to multiples of 8 (bottom / right) -> Conv2d(1, 16, 6, stride 2, pad 2) -> ReLU -> Conv2d(16, 32, 6, 2, 2) -> ReLU
//     -> Conv2d(32, 64, 6, 2, 2)  ->  [S, 64, H2/8, W2/8]   (the `proj` stack of FlowFormer's PatchEmbed for patch_size 8,
//   embed_dim = cost_latent_input_dim = 64; reached from Module/Network/FlowFormerCov/flownet.py:26 through MemoryEncoder; hyper-parameters
//   Config/Train/Demo.yaml:20-36; the result's token form [S, H2'·W2', C] is what covhead.py:61-64 documents for cost_memory's source).
//   The FlowFormer submodule is EMPTY in the reference checkout: layer shapes restated from the published FlowFormer sources — parity is
//   pinned to torch's F.conv2d chain on the same weights (oracle/patch_embed.py), "parity unpinned" against the MAC-VO fork itself.
//
// Why fused.  23.6 GFLOP of volume feed 2 x 9600 slices x 25.1 MFLOP = 241 GFLOP of convolutions per 640x480 frame — 10x the GEMM — and the
// unfused form moves every slice four times through HBM (read 184 MB, write + read 2 x 197 MB of 16-channel maps, 2 x 98 MB of 32-channel
// maps, write 197 MB).  Here a workgroup keeps a slice and both intermediate maps in LDS: HBM traffic is the slice in (19.2 KB) and the
// tokens out (20.5 KB fp32), every multiply-add runs on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, fp32 accumulate), activations are
// rounded to bf16 between the layers (the reference runs this encoder in fp16 / bf16 in its Fast mode, MACVO_Fast.yaml:73-74).
//
// Implicit GEMMs (M = output pixels, N = output channels, K = taps x input channels, cin innermost so that one lane's 8 consecutive k are 16
// contiguous bytes of an HWC activation map in LDS):
//   conv1  M = 32x40 = 1280 (80 tiles of 16 pixels, v_mfma_f32_16x16x32_bf16), N = 16, K = 8 ky x 8 kx' (taps >= 6 carry zero weights) = 2 k-steps
//   conv2  M = 16x20 = 320 (10 tiles), N = 32, K = 36 taps x 16 = 36 k-steps; waves = (K half, every other tile): 5 tiles x 18 k-steps each,
//          each wave keeps the weight fragments of its K half (18 KB) in registers for the whole kernel
//   conv3  TWO slices per pass: M = 2 x 8x10 = 160 (5 tiles), N = 64 (2 tiles), K = 36 taps x 32 = 72 k-steps; waves = (K half, N tile): 5 tiles x
//          36 k-steps each; its 144 KB of weights stream from L2 (one 1-KB fragment per wave and k-step feeds 5 MFMAs; two slices per pass halve
//          that stream: 73.7 KB per slice)
//   K halves are summed through LDS (the dead conv1 map), each wave of a pair finishing half of the outputs.  720 32x32x16 + 160 16x16x32 MFMAs per
//   slice = 6400 matrix-pipe cycles per wave and slice.
// LDS: slice (bf16, halo 2 (+2 rows / +4 columns for the padding taps), pitch 88) 12,320 + conv1 map 4 planes x 36 rows x 26 cells 59,968 + 2 x conv2 map
// 8 planes x 20 rows x 13 cells 33,408 = 139,104 B;
// the two maps are stored [8-channel chunk][column parity][row][column / 2][16 B] (see PE).
#include "common.h"
#include <algorithm>
#include <atomic>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

template <int H2, int W2>
struct PE {
    static constexpr int HP = (H2 + 7) / 8 * 8, WP = (W2 + 7) / 8 * 8;
    static constexpr int H1 = HP / 2, W1 = WP / 2, H2o = HP / 4, W2o = WP / 4, H3 = HP / 8, W3 = WP / 8;
    static constexpr int M1 = H1 * W1, M2 = H2o * W2o, M3 = H3 * W3;
    static constexpr int T1 = M1 / 16, T2 = M2 / 32, T3 = 2 * M3 / 32;     // conv1: 16-pixel tiles (16x16x32 MFMA); conv2 / conv3: 32-pixel tiles (conv3: of the two slices of a pass)
    static_assert(M1 % 64 == 0 && M2 % 64 == 0 && (2 * M3) % 32 == 0, "tile split across the four waves");
    static constexpr int IN_ROWS = HP + 6, IN_PITCH = WP + 8;              // halo 2; the zero-weight padding taps (ky, kx' = 6, 7) read two rows / columns further
    static constexpr int O1_ROWS = H1 + 4, O1_COLS = W1 + 4;               // x 16 channels (bf16)
    static constexpr int O2_ROWS = H2o + 4, O2_COLS = W2o + 4;             // x 32 channels
    // Activation maps in LDS are stored [8-channel chunk][column parity][row][column / 2][8 channels = 16 B] (round 4, second pass): a
    // stride-2 convolution's fragment read — 32 lanes = 32 consecutive output pixels, one tap, one chunk — then walks CONSECUTIVE 16-byte
    // cells of one parity plane (bank-conflict-free up to the row wrap) instead of cells 64 / 128 bytes apart (HWC: 4-way conflicts in conv2,
    // 8-way in conv3, measured as 85 % of the kernel's time being LDS-bound).  Same bytes, same sizes.
    static_assert(O1_COLS % 2 == 0 && O2_COLS % 2 == 0, "column-parity planes");
    // Row pitch of a plane in 16-byte cells.  A fragment read's 16-lane service groups span up to three output rows; the next output row
    // is 2 x pitch cells further, and the groups tile all 16 bank quads exactly when 2 x pitch = 4 (mod 16) for 20-pixel rows (conv2) and
    // = 10 (mod 16) for 10-pixel rows (conv3) — simulated over every tile alignment: 2.0 / 2.4 LDS cycles per 32 lanes against 3.6 / 5.6 for
    // the tight pitches 22 / 12 (the room comes from keeping conv2's weights in registers instead of LDS).
    static constexpr int pad_xh(int lo, int r) { int x = lo; while (x % 8 != r) ++x; return x; }
    static constexpr int O1_XH = pad_xh(O1_COLS / 2, 2), O2_XH = pad_xh(O2_COLS / 2, 5);
    // one (chunk, parity) plane, + one 16-byte cell: without it the planes of a pixel's 2 / 4 channel chunks start on the same bank and the
    // epilogues' ds_write_b16 (32 lanes = 32 channels of one pixel) were 8-way conflicted
    static constexpr unsigned O1_PLANE = O1_ROWS * O1_XH * 16 + 16, O2_PLANE = O2_ROWS * O2_XH * 16 + 16;
    // byte offset of 8-channel chunk c of padded cell (row, col)
    static constexpr unsigned o1_cell(int c, int row, int col) { return (unsigned)(c * 2 + (col & 1)) * O1_PLANE + (unsigned)(row * O1_XH + (col >> 1)) * 16; }
    static constexpr unsigned o2_cell(int c, int row, int col) { return (unsigned)(c * 2 + (col & 1)) * O2_PLANE + (unsigned)(row * O2_XH + (col >> 1)) * 16; }
    static constexpr unsigned OFF_IN0 = 0, IN0_BYTES = IN_ROWS * IN_PITCH * 2;
    static constexpr unsigned OFF_O1 = OFF_IN0 + IN0_BYTES, O1_BYTES = 4 * O1_PLANE;      // 2 chunks x 2 parity planes
    static constexpr unsigned OFF_O2 = OFF_O1 + O1_BYTES, O2_BYTES = 8 * O2_PLANE;      // 4 chunks x 2 parity planes
    static constexpr unsigned LDS_BYTES = OFF_O2 + 2 * O2_BYTES;
    static constexpr unsigned SCRATCH_BYTES = 2 * 5 * 16 * 256;            // K-half partial sums: 2 waves x 5 tiles x 16 registers x 64 lanes fp32
    static_assert(T2 == 10 && T3 == 5, "five tiles per wave in conv2 and conv3");
    static_assert(SCRATCH_BYTES <= O1_BYTES && LDS_BYTES <= 160 * 1024, "LDS plan");
    static constexpr int Q4 = H2 * W2 / 4;                                 // float4s of a slice
    static_assert(W2 % 4 == 0 && Q4 <= 5 * 256, "slice staging: at most five float4 per thread");
};

// packed weights (mv_patch_embed_pack): bf16 fragments in MFMA B-operand order — lane l holds n = l % 32, k = 8 (l / 32) + 0..7 — then biases
//   [0, 3 KB)            conv1 (16x16x32 fragments: lane l = channel l % 16): 2 k-steps; k = (ky = 4 s + l / 16, kx' = j), zero for ky, kx' >= 6
//   [3 KB, 39 KB)        conv2: k-step = tap ky*6 + kx; k = cin
//   [39 KB, 183 KB)      conv3: [n tile 2][k-step 72 = tap * 2 + cin half]; k = cin % 16
//   then fp32 b1[32] (16 used), b2[32], b3[64]
constexpr size_t PE_W1_OFF = 0, PE_W2_OFF = 3 * 1024, PE_W3_OFF = 39 * 1024, PE_B_OFF = 183 * 1024, PE_PACKED_BYTES = PE_B_OFF + 128 * 4;

__global__ void patch_embed_pack_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                        const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3,
                                        uint16_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // one 16-bit element of the fragment area
    const int total = (int)(PE_B_OFF / 2);
    if (i < total) {
        const int unit = i >> 9, lane = (i >> 3) & 63, j = i & 7, n = lane & 31, g = lane >> 5;
        float v = 0.f;
        if (unit < 3) {                                        // conv1 [16,1,6,6]: B operand of v_mfma_f32_16x16x32: lane l = channel l % 16, k = 8 (l / 16) + j
            const int n16 = lane & 15, ky = 4 * unit + (lane >> 4);      // k-step `unit` (2 used): k = (ky = 4 unit + l / 16, kx' = j); ky, kx' >= 6: zero
            if (unit < 2 && ky < 6 && j < 6) v = w1[(n16 * 6 + ky) * 6 + j];
        } else if (unit < 39) {                                // conv2 [32,16,6,6]
            const int tap = unit - 3, cin = g * 8 + j;
            v = w2[((size_t)n * 16 + cin) * 36 + tap];
        } else {                                               // conv3 [64,32,6,6]
            const int u = unit - 39, nt = u / 72, t = u % 72, tap = t >> 1, cin = (t & 1) * 16 + g * 8 + j;
            v = w3[((size_t)(nt * 32 + n) * 32 + cin) * 36 + tap];
        }
        out[i] = __builtin_bit_cast(uint16_t, (__bf16)v);
    }
    if (i < 128) {
        float* b = reinterpret_cast<float*>(reinterpret_cast<char*>(out) + PE_B_OFF);
        b[i] = i < 16 ? b1[i] : (i < 32 ? 0.f : (i < 64 ? b2[i - 32] : b3[i - 64]));
    }
}

__device__ __forceinline__ uint16_t bf16_bits(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }
__device__ __forceinline__ unsigned bf16_pack(float lo, float hi) { return (unsigned)bf16_bits(lo) | ((unsigned)bf16_bits(hi) << 16); }

template <int H2, int W2, bool TOKENS>
__global__ __launch_bounds__(256) void cost_patch_embed_kernel(const float* __restrict__ vol, const char* __restrict__ wp,
                                                               float* __restrict__ out, int S) {
    using P = PE<H2, W2>;
    extern __shared__ __attribute__((aligned(16))) char smem_pe[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n32 = lane & 31, g = lane >> 5;
    const int kh = wave & 1, hs = wave >> 1;           // K half; conv2: tile parity, conv3: N tile
    char* const in0 = smem_pe + P::OFF_IN0;
    char* const o1 = smem_pe + P::OFF_O1;
    char* const o2 = smem_pe + P::OFF_O2;

    // ---- once per workgroup: zero the activation buffers (their halos stay zero), conv2 weights -> LDS, conv1 weights + biases -> registers
    for (unsigned a = (unsigned)t * 16u; a < P::LDS_BYTES - P::OFF_IN0; a += 256u * 16u) *reinterpret_cast<i32x4*>(in0 + a) = i32x4{0, 0, 0, 0};
    // conv2's B fragments of this wave's K half stay in registers for the whole kernel (18 x 4 VGPRs): no LDS traffic for them, and the 36 KB
    // they used to occupy pay for the conflict-free row pitches above
    i32x4 w2f[18];
#pragma unroll
    for (int kk = 0; kk < 18; ++kk) w2f[kk] = *reinterpret_cast<const i32x4*>(wp + PE_W2_OFF + ((size_t)(kh * 18 + kk) * 64 + lane) * 16);
    i32x4 w1f[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) w1f[s] = *reinterpret_cast<const i32x4*>(wp + PE_W1_OFF + (s * 64 + lane) * 16);
    const float* bias = reinterpret_cast<const float*>(wp + PE_B_OFF);
    const float b1v = bias[lane & 15], b2v = bias[32 + n32], b3v = bias[64 + hs * 32 + n32];
    const char* const w3 = wp + PE_W3_OFF + ((size_t)(hs * 72 + kh * 36) * 64 + lane) * 16;   // this wave's 36 conv3 fragments
    __syncthreads();

    auto zero_o1_halo = [&]() {                        // the K-half scratch lives in the conv1 map: restore its zero halo afterwards
        for (int c = t; c < 4 * P::O1_COLS + 4 * P::H1; c += 256) {
            int row, col;
            if (c < 4 * P::O1_COLS) {
                const int r = c / P::O1_COLS;
                row = r < 2 ? r : P::O1_ROWS - 4 + r;
                col = c - r * P::O1_COLS;
            } else {
                const int d = c - 4 * P::O1_COLS, r = d >> 2, q = d & 3;
                row = 2 + r;
                col = q < 2 ? q : P::O1_COLS - 4 + q;
            }
            *reinterpret_cast<i32x4*>(o1 + P::o1_cell(0, row, col)) = i32x4{0, 0, 0, 0};
            *reinterpret_cast<i32x4*>(o1 + P::o1_cell(1, row, col)) = i32x4{0, 0, 0, 0};
        }
    };
    // slice staging: this thread's float4s of the NEXT slice travel in registers while the current one is convolved
    f32x4 pre[5];
    auto fetch = [&](int s) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int q = t + 256 * i;
            pre[i] = (s < S && q < P::Q4) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vol + (size_t)s * (H2 * W2)) + q) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // conv1 addressing (see the conv1 phase): fragment base in the slice and store base in the conv1 map for this wave's tiles 0..4
    static_assert(P::T1 % 20 == 0 && (20 * 16) % P::W1 == 0, "conv1: five tiles per wave span whole output rows");
    const char* c1_a[5];
    char* c1_d[5];
    {
        const int n16 = lane & 15, g4 = lane >> 4;
#pragma unroll
        for (int r5 = 0; r5 < 5; ++r5) {
            const int tile = wave + 4 * r5;
            const int p = tile * 16 + n16, oy = p / P::W1, ox = p - oy * P::W1;
            c1_a[r5] = in0 + ((2 * oy + g4) * P::IN_PITCH + 2 * ox) * 2;
            const int pp = tile * 16 + 4 * g4, y = pp / P::W1, x = pp - y * P::W1;      // four consecutive pixels of one row (W1 % 4 == 0)
            c1_d[r5] = o1 + (n16 >> 3) * 2 * P::O1_PLANE + (n16 & 7) * 2 + P::o1_cell(0, y + 2, x + 2);
        }
    }
    const int npass = (S + 1) >> 1;
    int pass = blockIdx.x;
    if (pass < npass) fetch(2 * pass);
    for (; pass < npass; pass += gridDim.x) {
#pragma unroll 1
        for (int gs = 0; gs < 2; ++gs) {
            int gv = g;                                 // the epilogues' address arithmetic starts from this copy: opaque per iteration, so that hipcc
            asm volatile("" : "+v"(gv));                // does not hoist ~100 loop-invariant addresses out of the slice loop (it spilled 131 registers)
            // ---- (A) slice -> in0 (bf16), next slice -> registers
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int q = t + 256 * i;
                if (q < P::Q4) {
                    const int y = q / (W2 / 4), x = 4 * (q - y * (W2 / 4));
                    unsigned* d = reinterpret_cast<unsigned*>(in0 + ((y + 2) * P::IN_PITCH + x + 2) * 2);
                    d[0] = bf16_pack(pre[i][0], pre[i][1]);
                    d[1] = bf16_pack(pre[i][2], pre[i][3]);
                }
            }
            fetch(gs == 0 ? 2 * pass + 1 : 2 * (pass + (int)gridDim.x));
            __syncthreads();
            // ---- (C) conv1: 10 tiles per wave, 3 k-steps; k-step s of lane group g = input row 2 oy + 2 s + g, columns 2 ox .. 2 ox + 7
            {
                // v_mfma_f32_16x16x32_bf16: tile = 16 consecutive output pixels x 16 channels, K = 2 x 32 = (ky 0..7) x (kx' 0..7), taps >= 6 carry zero
                // weights.  Every lane ends with FOUR consecutive pixels of ONE channel (C layout: column = lane % 16, rows 4 (lane / 16) + 0..3):
                // all 64 lanes take part in the epilogue (the 32x32 form used half of them and twice the outputs per lane).
                // Tile j of this wave = pixels (wave + 4 j) 16 ..: 20 tiles = 8 output rows (320 pixels) further per 5 tiles, so the fragment and the
                // store address of tile j = 5 m + r are those of tile r plus m constant strides: ten addresses computed ONCE per kernel (c1_a / c1_d)
                // replace two divisions by W1 per tile — the conv1 phase was ~900 VALU instructions per slice for 40 MFMAs.
#pragma unroll
                for (int m = 0; m < P::T1 / 20; ++m)
#pragma unroll
                    for (int r5 = 0; r5 < 5; ++r5) {
                        const char* a0 = c1_a[r5] + m * (16 * P::IN_PITCH * 2);
                        i32x4 af[2];
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            const unsigned* ap = reinterpret_cast<const unsigned*>(a0 + 4 * s * P::IN_PITCH * 2);
                            af[s] = i32x4{(int)ap[0], (int)ap[1], (int)ap[2], (int)ap[3]};
                        }
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int s = 0; s < 2; ++s)
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[s]), __builtin_bit_cast(bf16x8, w1f[s]), acc, 0, 0, 0);
                        char* d = c1_d[r5] + m * (8 * P::O1_XH * 16);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            *reinterpret_cast<uint16_t*>(d + (e & 1) * P::O1_PLANE + (e >> 1) * 16) = bf16_bits(fmaxf(acc[e] + b1v, 0.f));
                    }
            }
            __syncthreads();
            // ---- (D) conv2: this wave = K half kh (taps 18 kh ..), tiles hs, hs + 2, .. (5)
            {
                f32x16 acc[5];
                const char* abase[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
                    const int p = (hs + 2 * i) * 32 + n32, oy = p / P::W2o, ox = p - oy * P::W2o;
                    abase[i] = o1 + P::o1_cell(g, 2 * oy, 2 * ox);       // chunk g of padded cell (2 oy, 2 ox); tap (ky, kx) = cell (2 oy + ky, 2 ox + kx)
                }
                auto taps = [&](auto KH) __attribute__((always_inline)) {   // (kh is wave-uniform: one instantiation per K half keeps every offset an immediate)
                    constexpr int K0 = decltype(KH)::value * 18, PFA = 2;        // fragments are fetched PFA k-steps ahead of their MFMAs
                    bf16x8 af[PFA + 1][5];
                    auto fetch_k = [&](int kk) __attribute__((always_inline)) {
                        const int tap = K0 + kk, ky = tap / 6, kx = tap - 6 * ky, q = kk % (PFA + 1);
#pragma unroll
                        for (int i = 0; i < 5; ++i) af[q][i] = *reinterpret_cast<const bf16x8*>(abase[i] + P::o1_cell(0, ky, kx));
                    };
#pragma unroll
                    for (int kk = 0; kk < PFA; ++kk) fetch_k(kk);
#pragma unroll
                    for (int kk = 0; kk < 18; ++kk) {
                        if (kk + PFA < 18) fetch_k(kk + PFA);
                        __builtin_amdgcn_sched_barrier(0);                       // keep the prefetch ahead of the MFMAs (hipcc sinks it back otherwise)
#pragma unroll
                        for (int i = 0; i < 5; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk % (PFA + 1)][i], __builtin_bit_cast(bf16x8, w2f[kk]), acc[i], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                if (kh == 0) taps(std::integral_constant<int, 0>{});
                else taps(std::integral_constant<int, 1>{});
                __syncthreads();                                        // everyone has read the conv1 map: it becomes the K-half scratch
                // K halves summed through LDS, SYMMETRICALLY: of a wave pair's 80 (tile, register) values the kh = 0 wave finishes the first 40 and the
                // kh = 1 wave the last 40; each hands the other 40 partial sums (slot = index % 40 of the writer's region) — no wave idles through
                // the other's epilogue (the one-sided form had two waves waiting for ~480 VALU instructions of the other two).
                float* const scw = reinterpret_cast<float*>(o1) + (size_t)(hs * 2 + kh) * (40 * 64) + lane;         // this wave writes here
                const float* const scr = reinterpret_cast<const float*>(o1) + (size_t)(hs * 2 + (kh ^ 1)) * (40 * 64) + lane;   // ... and reads its partner's
                auto hand_over = [&](auto KH) __attribute__((always_inline)) {
#pragma unroll
                    for (int i = 0; i < 5; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            constexpr int MINE_LO = decltype(KH)::value * 40;
                            const int idx = i * 16 + r;
                            if (idx < MINE_LO || idx >= MINE_LO + 40) scw[(idx % 40) * 64] = acc[i][r];
                        }
                };
                if (kh == 0) hand_over(std::integral_constant<int, 0>{});
                else hand_over(std::integral_constant<int, 1>{});
                __syncthreads();
                char* o2g = o2 + gs * P::O2_BYTES;
                auto finish = [&](auto KH) __attribute__((always_inline)) {
#pragma unroll
                    for (int i = 0; i < 5; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            constexpr int MINE_LO = decltype(KH)::value * 40;
                            if (i * 16 + 4 * q < MINE_LO || i * 16 + 4 * q >= MINE_LO + 40) continue;       // (quads do not straddle the split: 40 % 4 == 0)
                            const int pp = (hs + 2 * i) * 32 + 8 * q + 4 * gv, y = pp / P::W2o, x = pp - y * P::W2o;
                            char* d = o2g + P::o2_cell(n32 >> 3, y + 2, x + 2) + (n32 & 7) * 2;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int r = 4 * q + e, idx = i * 16 + r;
                                *reinterpret_cast<uint16_t*>(d + (e & 1) * P::O2_PLANE + (e >> 1) * 16) =
                                    bf16_bits(fmaxf(acc[i][r] + scr[(idx % 40) * 64] + b2v, 0.f));
                            }
                        }
                };
                if (kh == 0) finish(std::integral_constant<int, 0>{});
                else finish(std::integral_constant<int, 1>{});
                __syncthreads();
                zero_o1_halo();
            }
        }
        // ---- (E) conv3 over the two slices of the pass: this wave = K half kh (k-steps 36 kh ..), N tile hs; weights stream from L2
        {
            int gv = g;
            asm volatile("" : "+v"(gv));
            f32x16 acc[5];
            const char* abase[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
                const int p = i * 32 + n32, sl = p / P::M3, q = p - sl * P::M3, oy = q / P::W3, ox = q - oy * P::W3;
                abase[i] = o2 + sl * P::O2_BYTES + P::o2_cell(g, 2 * oy, 2 * ox);   // chunk (2 hh + g) of cell (2 oy + ky, 2 ox + kx) per k-step
            }
            constexpr int PF = 6;                                       // weight fragments in flight (L2 latency / 5 MFMAs per k-step)
            i32x4 bq[PF];
#pragma unroll
            for (int j = 0; j < PF; ++j) bq[j] = *reinterpret_cast<const i32x4*>(w3 + j * 1024);
            auto ksteps = [&](auto KH) __attribute__((always_inline)) {
                constexpr int K0 = decltype(KH)::value * 36, PFA = 2;
                bf16x8 af[PFA + 1][5];
                auto fetch_a = [&](int kk) __attribute__((always_inline)) {
                    const int tt = K0 + kk, tap = tt >> 1, ky = tap / 6, kx = tap - 6 * ky, hh = tt & 1;
#pragma unroll
                    for (int i = 0; i < 5; ++i) af[kk % (PFA + 1)][i] = *reinterpret_cast<const bf16x8*>(abase[i] + P::o2_cell(2 * hh, ky, kx));
                };
#pragma unroll
                for (int kk = 0; kk < PFA; ++kk) fetch_a(kk);
#pragma unroll
                for (int kk = 0; kk < 36; ++kk) {
                    const i32x4 b = bq[kk % PF];
                    if (kk + PFA < 36) fetch_a(kk + PFA);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 5; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk % (PFA + 1)][i], __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
                    if (kk + PF < 36) bq[kk % PF] = *reinterpret_cast<const i32x4*>(w3 + (kk + PF) * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (kh == 0) ksteps(std::integral_constant<int, 0>{});
            else ksteps(std::integral_constant<int, 1>{});
            __syncthreads();                                            // (the halo writes of (D) are complete in every wave)
            float* const scw = reinterpret_cast<float*>(o1) + (size_t)(hs * 2 + kh) * (40 * 64) + lane;          // symmetric hand-over as in conv2
            const float* const scr = reinterpret_cast<const float*>(o1) + (size_t)(hs * 2 + (kh ^ 1)) * (40 * 64) + lane;
            auto hand_over = [&](auto KH) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        constexpr int MINE_LO = decltype(KH)::value * 40;
                        const int idx = i * 16 + r;
                        if (idx < MINE_LO || idx >= MINE_LO + 40) scw[(idx % 40) * 64] = acc[i][r];
                    }
            };
            if (kh == 0) hand_over(std::integral_constant<int, 0>{});
            else hand_over(std::integral_constant<int, 1>{});
            __syncthreads();
            const int ch = hs * 32 + n32;
            float* const opass = out + (size_t)2 * pass * (P::M3 * 64);           // (uniform) first token of the pass's first slice
            auto finish = [&](auto KH) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        constexpr int MINE_LO = decltype(KH)::value * 40;
                        if (i * 16 + 4 * qd < MINE_LO || i * 16 + 4 * qd >= MINE_LO + 40) continue;
                        const int pp = i * 32 + 8 * qd + 4 * gv, sl = pp / P::M3, q = pp - sl * P::M3;    // four consecutive tokens of one slice (M3 % 4 == 0)
                        if (2 * pass + sl < S) {
                            float* d = opass + (TOKENS ? (sl * P::M3 + q) * 64 + ch : (sl * 64 + ch) * P::M3 + q);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int r = 4 * qd + e, idx = i * 16 + r;
                                __builtin_nontemporal_store(acc[i][r] + scr[(idx % 40) * 64] + b3v, d + (TOKENS ? e * 64 : e));
                            }
                        }
                    }
            };
            if (kh == 0) finish(std::integral_constant<int, 0>{});
            else finish(std::integral_constant<int, 1>{});
            __syncthreads();
            zero_o1_halo();
        }
    }
}

}  // namespace

extern "C" size_t mv_patch_embed_packed_bytes(void) { return PE_PACKED_BYTES; }

extern "C" int mv_patch_embed_pack(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                                   void* packed, mvStream_t stream) {
    MV_CHECK_ARG(w1 && b1 && w2 && b2 && w3 && b3 && packed && ((uintptr_t)packed & 15) == 0);
    const int total = (int)(PE_B_OFF / 2);
    hipLaunchKernelGGL(patch_embed_pack_kernel, dim3(mv_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w1, b1, w2, b2, w3, b3,
                       (uint16_t*)packed);
    return mv_launch_status();
}

extern "C" int mv_cost_patch_embed_supported(int H2, int W2) { return H2 == 60 && W2 == 80; }

extern "C" int mv_cost_patch_embed(const float* cost_maps, const void* packed, float* out, int S, int H2, int W2, int token_layout,
                                   mvStream_t stream) {
    MV_CHECK_ARG(cost_maps && packed && out && S > 0);
    MV_CHECK_ARG(((uintptr_t)cost_maps & 15) == 0 && ((uintptr_t)packed & 15) == 0);
    if (!mv_cost_patch_embed_supported(H2, W2)) return MV_ERR_UNSUPPORTED;   // 640x480 frames; larger slices do not fit the LDS plan
    using P = PE<60, 80>;
    static std::atomic<bool> attr_done[64];
    static int cus = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[dev].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute((const void*)cost_patch_embed_kernel<60, 80, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)cost_patch_embed_kernel<60, 80, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done[dev].store(true, std::memory_order_release);
    }
    if (!cus) {
        hipDeviceProp_t prop;
        cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    const int npass = (S + 1) / 2;
    const dim3 grid(std::min(npass, cus));                                  // persistent: one workgroup per CU (160,960 B of LDS each)
    if (token_layout)
        hipLaunchKernelGGL((cost_patch_embed_kernel<60, 80, true>), grid, dim3(256), P::LDS_BYTES, (hipStream_t)stream, cost_maps,
                           (const char*)packed, out, S);
    else
        hipLaunchKernelGGL((cost_patch_embed_kernel<60, 80, false>), grid, dim3(256), P::LDS_BYTES, (hipStream_t)stream, cost_maps,
                           (const char*)packed, out, S);
    return mv_launch_status();
}
