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
#include <stdlib.h>
#include <type_traits>

#include "patch_embed_dev.h"
using namespace pe;

namespace {


__global__ void patch_embed_pack_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                        const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3,
                                        uint16_t* __restrict__ out, int f16) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // one 16-bit element of the fragment area
    const int total = (int)(PE_B_OFF / 2);
    if (i < total) {
        const int unit = i >> 9, lane = (i >> 3) & 63, j = i & 7;
        float v = 0.f;
        if (unit < 3) {                                        // conv1 [16,1,6,6]: B operand of v_mfma_f32_16x16x32: lane l = channel l % 16, k = 8 (l / 16) + j
            // k-step `unit` (2 used): k = (ky, kx' = j) with ky = 4 (g & 1) + (g >> 1) + 2 unit for lane group g = l / 16 (round 5; was 4 unit + g): the two
            // lane groups that share a ds_read_b32 service group (g = 0, 1 / 2, 3) then read input rows FOUR apart = 16 banks apart for every row pitch with
            // pitch / 2 = 4 (mod 8) dwords, instead of neighbouring rows 12 banks apart (2-way conflicts on every conv1 fragment read); ky, kx' >= 6: zero
            const int n16 = lane & 15, g = lane >> 4, ky = 4 * (g & 1) + (g >> 1) + 2 * unit;
            if (unit < 2 && ky < 6 && j < 6) v = w1[(n16 * 6 + ky) * 6 + j];
        } else if (unit < 39) {                                // conv2 [32,16,6,6]: unit = 3 + nt * 18 + ks; lane = channel nt * 16 + l % 16; k = 8 (l / 16) + j
            const int u = unit - 3, nt = u / 18, ks = u - nt * 18, n16 = lane & 15, g4 = lane >> 4;
            const int tap = 2 * ks + (g4 >> 1), cin = (g4 & 1) * 8 + j;        // a k-step = two neighbouring taps x 16 input channels
            v = w2[((size_t)(nt * 16 + n16) * 16 + cin) * 36 + tap];
        } else {                                               // conv3 [64,32,6,6]: unit = 39 + nt * 36 + tap; k = input channel 8 (l / 16) + j
            const int u = unit - 39, nt = u / 36, tap = u - nt * 36, n16 = lane & 15, cin = (lane >> 4) * 8 + j;
            v = w3[((size_t)(nt * 16 + n16) * 32 + cin) * 36 + tap];
        }
        out[i] = f16 ? __builtin_bit_cast(uint16_t, (_Float16)v) : __builtin_bit_cast(uint16_t, (__bf16)v);
    }
    if (i < 128) {
        float* b = reinterpret_cast<float*>(reinterpret_cast<char*>(out) + PE_B_OFF);
        b[i] = i < 16 ? b1[i] : (i < 32 ? 0.f : (i < 64 ? b2[i - 32] : b3[i - 64]));
    }
}

#ifdef MV_PE_PHASE_REFERENCE   // the phase-by-phase kernel: TEST-ONLY build (tests/pe_phase_ref.py), the bitwise reference of patch_embed_v3.hip
// IN16 / OUT16: the slice is read / the tokens are written in the OPERAND type (fp16 cells of the `out16` volume -> fp16 tokens for the fp16 encoder of
// MACVO_Fast.yaml:73-74; bf16 likewise) instead of fp32: the cells go to LDS as they are, no conversion, and HBM traffic is 9.6 + 10.2 KB per
// slice instead of 19.2 + 20.5 KB
template <int H2, int W2, bool TOKENS, bool F16, bool IN16, bool OUT16>
__global__ __launch_bounds__(256) void cost_patch_embed_kernel(const void* __restrict__ vol_, const char* __restrict__ wp,
                                                               void* __restrict__ out_, int S) {
    using P = PE<H2, W2>;
    constexpr int EPL = IN16 ? 8 : 4;                                       // cells per 16-byte load
    constexpr int QN = H2 * W2 / EPL, NPRE = (QN + 255) / 256;
    static_assert(W2 % EPL == 0 && NPRE <= 5, "slice staging");
    const char* const vol = reinterpret_cast<const char*>(vol_);
    extern __shared__ __attribute__((aligned(16))) char smem_pe[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n16 = lane & 15, g4 = lane >> 4;
    const int mh = wave & 1, np = wave >> 1;           // conv3: slice of the pass, pair of 16-channel tiles
    char* const in0 = smem_pe + P::OFF_IN0;
    char* const o1 = smem_pe + P::OFF_O1;
    char* const o2 = smem_pe + P::OFF_O2;

    // ---- once per workgroup: zero the activation buffers (their halos stay zero); conv1 / conv2 weights + biases -> registers
    for (unsigned a = (unsigned)t * 16u; a < P::OFF_W2B; a += 256u * 16u) *reinterpret_cast<i32x4*>(smem_pe + a) = i32x4{0, 0, 0, 0};
    // conv2's B fragments: channel tile 0 (18 k-steps x 4 VGPRs) stays in registers for the whole kernel, tile 1 in LDS (all 36 in registers spilled;
    // all 36 in LDS cost the 36 KB that now pay for the conflict-free row pitches above)
    i32x4 w2f[18];
#pragma unroll
    for (int ks = 0; ks < 18; ++ks) w2f[ks] = *reinterpret_cast<const i32x4*>(wp + PE_W2_OFF + ((size_t)ks * 64 + lane) * 16);
    for (unsigned a = (unsigned)t * 16u; a < P::W2B_BYTES; a += 256u * 16u)
        *reinterpret_cast<i32x4*>(smem_pe + P::OFF_W2B + a) = *reinterpret_cast<const i32x4*>(wp + PE_W2_OFF + 18 * 1024 + a);
    i32x4 w1f[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) w1f[s] = *reinterpret_cast<const i32x4*>(wp + PE_W1_OFF + (s * 64 + lane) * 16);
    const float* bias = reinterpret_cast<const float*>(wp + PE_B_OFF);
    const float b1v = bias[n16];
    const float b2v[2] = {bias[32 + n16], bias[48 + n16]};
    const float b3v[2] = {bias[64 + np * 32 + n16], bias[80 + np * 32 + n16]};
    const char* const w3 = wp + PE_W3_OFF + ((size_t)(2 * np) * 36 * 64 + lane) * 16;   // this wave's 2 x 36 conv3 fragments: (j, tap) at + (j * 36 + tap) KB
    __syncthreads();

    // slice staging: this thread's float4s of the NEXT slice travel in registers while the current one is convolved
    i32x4 pre[NPRE];
    auto fetch = [&](int s) {
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int q = t + 256 * i;
            pre[i] = (s < S && q < QN) ? __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(vol + (size_t)s * (H2 * W2) * (IN16 ? 2 : 4)) + q) : i32x4{0, 0, 0, 0};
        }
    };
    // conv1 addressing (see the conv1 phase): fragment base in the slice and store base in the conv1 map for this wave's tiles 0..4
    static_assert(P::T1 % 20 == 0 && (20 * 16) % P::W1 == 0, "conv1: five tiles per wave span whole output rows");
    const char* c1_a[5];
    char* c1_d[5];
    {
#pragma unroll
        for (int r5 = 0; r5 < 5; ++r5) {
            const int tile = wave + 4 * r5;
            const int p = tile * 16 + n16, oy = p / P::W1, ox = p - oy * P::W1;
            c1_a[r5] = in0 + ((2 * oy + 4 * (g4 & 1) + (g4 >> 1)) * P::IN_PITCH + 2 * ox) * 2;   // ky of k-step 0 (see mv_patch_embed_pack)
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
            int gv = g4;                                // the epilogues' address arithmetic starts from this copy: opaque per iteration, so that hipcc
            asm volatile("" : "+v"(gv));                // does not hoist ~100 loop-invariant addresses out of the slice loop (it spilled 131 registers)
            // ---- (A) slice -> in0 (bf16), next slice -> registers
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const int q = t + 256 * i;
                if (q < QN) {
                    const int y = q / (W2 / EPL), x = EPL * (q - y * (W2 / EPL));
                    unsigned* d = reinterpret_cast<unsigned*>(in0 + ((y + 2) * P::IN_PITCH + x + 2) * 2);
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
                            const unsigned* ap = reinterpret_cast<const unsigned*>(a0 + 2 * s * P::IN_PITCH * 2);
                            af[s] = i32x4{(int)ap[0], (int)ap[1], (int)ap[2], (int)ap[3]};
                        }
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int s = 0; s < 2; ++s)
                            acc = mma16<F16>(__builtin_bit_cast(bf16x8, af[s]), __builtin_bit_cast(bf16x8, w1f[s]), acc);
                        char* d = c1_d[r5] + m * (8 * P::O1_XH * 16);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            *reinterpret_cast<uint16_t*>(d + (e & 1) * P::O1_PLANE + (e >> 1) * 16) = cvt_bits<F16>(fmaxf(acc[e] + b1v, 0.f));
                    }
            }
            __syncthreads();
            // ---- (D) conv2 (v_mfma_f32_16x16x32_bf16): this wave = output rows 4 wave .. 4 wave + 3 (five 16-pixel tiles) x both 16-channel tiles; a k-step =
            // two neighbouring taps (2 ks, 2 ks + 1: same ky, kx = 2 (ks % 3) + h) x 16 input channels: lane group l / 16 = (h, channel chunk) reads chunk
            // (l / 16) % 2 of the parity-h plane, so the tap offset is the same immediate for every lane.  No K split: nothing to sum across waves.
            {
                f32x4 acc[5][2];
                const char* abase[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const int p = (5 * wave + i) * 16 + n16, oy = p / P::W2o, ox = p - oy * P::W2o;
                    abase[i] = o1 + P::o1_cell(g4 & 1, 2 * oy, 2 * ox) + (g4 >> 1) * P::O1_PLANE;
                }
                constexpr int PFA = 2;                                  // fragments are fetched PFA k-steps ahead of their MFMAs
                bf16x8 af[PFA + 1][5], bf1[PFA + 1];
                const char* const wb1 = smem_pe + P::OFF_W2B + lane * 16;
                auto fetch_k = [&](int ks) __attribute__((always_inline)) {
                    bf1[ks % (PFA + 1)] = *reinterpret_cast<const bf16x8*>(wb1 + ks * 1024);
#pragma unroll
                    for (int i = 0; i < 5; ++i)
                        af[ks % (PFA + 1)][i] = *reinterpret_cast<const bf16x8*>(abase[i] + ((ks / 3) * P::O1_XH + ks % 3) * 16);
                };
#pragma unroll
                for (int ks = 0; ks < PFA; ++ks) fetch_k(ks);
#pragma unroll
                for (int ks = 0; ks < 18; ++ks) {
                    if (ks + PFA < 18) fetch_k(ks + PFA);
                    __builtin_amdgcn_sched_barrier(0);                  // keep the prefetch ahead of the MFMAs (hipcc sinks it back otherwise)
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        acc[i][0] = mma16<F16>(af[ks % (PFA + 1)][i], __builtin_bit_cast(bf16x8, w2f[ks]), acc[i][0]);
                        acc[i][1] = mma16<F16>(af[ks % (PFA + 1)][i], bf1[ks % (PFA + 1)], acc[i][1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // epilogue: a lane holds four consecutive pixels (rows 4 (l / 16) + 0..3 of the tile) of channel nt * 16 + l % 16
                char* const o2g = o2 + gs * P::O2_BYTES + (n16 >> 3) * 2 * P::O2_PLANE + (n16 & 7) * 2;
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const int pp = (5 * wave + i) * 16 + 4 * gv, y = pp / P::W2o, x = pp - y * P::W2o;
                    char* d = o2g + P::o2_cell(0, y + 2, x + 2);
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            *reinterpret_cast<uint16_t*>(d + (nt * 4 + (e & 1)) * P::O2_PLANE + (e >> 1) * 16) = cvt_bits<F16>(fmaxf(acc[i][nt][e] + b2v[nt], 0.f));
                }
            }
        }
        __syncthreads();                                                // both conv2 maps of the pass are complete
        // ---- (E) conv3 over the two slices of the pass: this wave = slice mh (its 80 tokens = five 16-pixel tiles) x channel tiles 2 np, 2 np + 1; a k-step =
        // one tap x 32 input channels (lane group l / 16 = channel chunk); 2 x 36 weight fragments per wave and pass stream from L2
        {
            int gv = g4;
            asm volatile("" : "+v"(gv));
            f32x4 acc[5][2];
            const char* abase[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int q = i * 16 + n16, oy = q / P::W3, ox = q - oy * P::W3;
                abase[i] = o2 + mh * P::O2_BYTES + P::o2_cell(g4, 2 * oy, 2 * ox);
            }
            constexpr int PF = 8, PFA = 2;                              // weight fragment pairs in flight (L2 latency); activation fragments ahead
            i32x4 bq[PF][2];
#pragma unroll
            for (int kk = 0; kk < PF; ++kk)
#pragma unroll
                for (int j = 0; j < 2; ++j) bq[kk][j] = *reinterpret_cast<const i32x4*>(w3 + (size_t)(j * 36 + kk) * 1024);
            bf16x8 af[PFA + 1][5];
            auto fetch_a = [&](int tap) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 5; ++i) af[tap % (PFA + 1)][i] = *reinterpret_cast<const bf16x8*>(abase[i] + P::o2_cell(0, tap / 6, tap % 6));
            };
#pragma unroll
            for (int kk = 0; kk < PFA; ++kk) fetch_a(kk);
#pragma unroll
            for (int kk = 0; kk < 36; ++kk) {
                const i32x4 b0 = bq[kk % PF][0], b1 = bq[kk % PF][1];
                if (kk + PFA < 36) fetch_a(kk + PFA);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    acc[i][0] = mma16<F16>(af[kk % (PFA + 1)][i], __builtin_bit_cast(bf16x8, b0), acc[i][0]);
                    acc[i][1] = mma16<F16>(af[kk % (PFA + 1)][i], __builtin_bit_cast(bf16x8, b1), acc[i][1]);
                }
                if (kk + PF < 36) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) bq[kk % PF][j] = *reinterpret_cast<const i32x4*>(w3 + (size_t)(j * 36 + kk + PF) * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const int s_out = 2 * pass + mh;
            if constexpr (OUT16) {
                // 16-bit tokens.  Channel-major: four consecutive tokens of one channel = one 8-byte store.  Token-major: neighbouring lanes (channels
                // ch, ch + 1) swap half of their four tokens so that every lane writes two dwords of two channels each (8 lanes = 32 contiguous bytes
                // of a token, as many as the fp32 form's 16-lane groups cover)
                uint16_t* const oslice = reinterpret_cast<uint16_t*>(out_) + (size_t)s_out * (P::M3 * 64);
                const bool odd = n16 & 1;
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const int q = i * 16 + 4 * gv;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int ch = (2 * np + j) * 16 + n16;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + b3v[j];
                        if constexpr (TOKENS) {
                            const float r0 = __shfl_xor(odd ? v[0] : v[2], 1, 64), r1 = __shfl_xor(odd ? v[1] : v[3], 1, 64);
                            if (s_out < S) {
                                unsigned* d = reinterpret_cast<unsigned*>(oslice + (q + (odd ? 2 : 0)) * 64 + (ch & ~1));
                                d[0] = odd ? cvt_pack<F16>(r0, v[2]) : cvt_pack<F16>(v[0], r0);
                                d[32] = odd ? cvt_pack<F16>(r1, v[3]) : cvt_pack<F16>(v[1], r1);
                            }
                        } else if (s_out < S) {
                            unsigned* d = reinterpret_cast<unsigned*>(oslice + ch * P::M3 + q);
                            d[0] = cvt_pack<F16>(v[0], v[1]);
                            d[1] = cvt_pack<F16>(v[2], v[3]);
                        }
                    }
                }
            } else if (s_out < S) {
                float* const oslice = reinterpret_cast<float*>(out_) + (size_t)s_out * (P::M3 * 64);
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const int q = i * 16 + 4 * gv;                           // four consecutive tokens
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int ch = (2 * np + j) * 16 + n16;
                        // plain stores: a wave writes 64-byte halves of a token's 256-byte row (16 channels); with nt stores the halves went to HBM
                        // separately (WRITE_SIZE 258 MB for 197 MB); the L2 merges them
                        float* d = oslice + (TOKENS ? q * 64 + ch : ch * P::M3 + q);
                        if (TOKENS) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) d[e * 64] = acc[i][j][e] + b3v[j];
                        } else {
                            *reinterpret_cast<f32x4*>(d) = f32x4{acc[i][j][0] + b3v[j], acc[i][j][1] + b3v[j], acc[i][j][2] + b3v[j], acc[i][j][3] + b3v[j]};
                        }
                    }
                }
            }
        }
    }
}

#endif

}  // namespace

extern "C" size_t mv_patch_embed_packed_bytes(void) { return PE_PACKED_BYTES; }

extern "C" int mv_patch_embed_pack(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                                   void* packed, int operand_type, mvStream_t stream) {
    MV_CHECK_ARG(w1 && b1 && w2 && b2 && w3 && b3 && packed && ((uintptr_t)packed & 15) == 0);
    MV_CHECK_ARG(operand_type == MV_F16 || operand_type == MV_BF16);
    const int total = (int)(PE_B_OFF / 2);
    hipLaunchKernelGGL(patch_embed_pack_kernel, dim3(mv_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w1, b1, w2, b2, w3, b3,
                       (uint16_t*)packed, operand_type == MV_F16 ? 1 : 0);
    return mv_launch_status();
}

#ifndef MV_PE_PHASE_REFERENCE
// patch_embed_v2.hip: the strip-mined kernel for the other slice sizes
int mv_cost_patch_embed_strip(const void* cost_maps, int in16, const void* packed, void* out, int out16, int S, int H2, int W2, int token_layout, int f16,
                              mvStream_t stream);
int mv_cost_patch_embed_strip_supported(int H2, int W2);
// patch_embed_v3.hip: the whole-slice LDS plan, pipelined across slices by two wave groups
int mv_cost_patch_embed_pipelined(const void* cost_maps, int in16, const void* packed, void* out, int out16, int S, int H2, int W2, int token_layout, int f16,
                                  mvStream_t stream);

static bool whole_slice_plan(int H2, int W2) { return (H2 == 60 || H2 == 64) && W2 == 80; }   // patch_embed_v3.hip: two whole slices in LDS

extern "C" int mv_cost_patch_embed_supported(int H2, int W2) { return whole_slice_plan(H2, W2) || mv_cost_patch_embed_strip_supported(H2, W2); }

extern "C" int mv_cost_patch_embed_t(const void* cost_maps, int in_dtype, const void* packed, void* out, int out_dtype, int S, int H2, int W2,
                                     int token_layout, int operand_type, mvStream_t stream) {
    MV_CHECK_ARG(cost_maps && packed && out && S > 0);
    MV_CHECK_ARG(((uintptr_t)cost_maps & 15) == 0 && ((uintptr_t)packed & 15) == 0 && ((uintptr_t)out & 15) == 0);
    MV_CHECK_ARG(operand_type == MV_F16 || operand_type == MV_BF16);        // must be the type `packed` was built for
    // a 16-bit slice / token type is the operand type itself (the cells go to the matrix pipe as they are)
    MV_CHECK_ARG(in_dtype == MV_F32 || in_dtype == operand_type);
    MV_CHECK_ARG(out_dtype == MV_F32 || out_dtype == operand_type);
    // 640x480 frames: 60 x 80 slices (padded to 64 rows inside the kernel) or the already padded 64 x 80 slices PatchEmbed.forward hands to `proj`
    // (also 640x512 frames): the pipelined whole-slice kernel; every other supported size: the strip-mined kernel
    if (!mv_cost_patch_embed_supported(H2, W2)) return MV_ERR_UNSUPPORTED;
    if (in_dtype == MV_F32 && out_dtype != MV_F32) return MV_ERR_UNSUPPORTED;   // fp32 volume -> 16-bit tokens: no caller (the volume hook returns the encoder dtype)
    const bool f16 = operand_type == MV_F16, in16 = in_dtype != MV_F32, out16 = out_dtype != MV_F32;
    const char* pe_env = getenv("MV_PE_STRIP");                       // A/B (read per call: tests toggle it): 1 = the strip-mined kernel also where the
    const bool strip_all = pe_env && atoi(pe_env) == 1;               // whole-slice plan exists
    if (!whole_slice_plan(H2, W2) || strip_all)
        return mv_cost_patch_embed_strip(cost_maps, in16, packed, out, out16, S, H2, W2, token_layout, f16, stream);
    // (round 6: the phase-by-phase kernel this file used to launch under MV_PE_PIPELINED=0 is a test-only build now — -DMV_PE_PHASE_REFERENCE, tests/pe_phase_ref.py —
    // where it remains the bitwise reference of the pipelined kernel)
    return mv_cost_patch_embed_pipelined(cost_maps, in16, packed, out, out16, S, H2, W2, token_layout, f16, stream);
}

#else   // MV_PE_PHASE_REFERENCE: the test-only library exports the pack entry points above and this one launch
template <int H2, bool F16, bool IN16, bool OUT16>
static int launch_patch_embed(const void* cost_maps, const void* packed, void* out, int S, int token_layout, hipStream_t stream) {
    using P = PE<H2, 80>;
    static std::atomic<bool> attr_done[64];
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[dev].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute((const void*)cost_patch_embed_kernel<H2, 80, true, F16, IN16, OUT16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)cost_patch_embed_kernel<H2, 80, false, F16, IN16, OUT16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done[dev].store(true, std::memory_order_release);
    }
    int ncu = cus[dev].load(std::memory_order_relaxed);
    if (!ncu) {
        hipDeviceProp_t prop;
        ncu = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        cus[dev].store(ncu, std::memory_order_relaxed);
    }
    const int npass = (S + 1) / 2;
    const dim3 grid(std::min(npass, ncu));                                  // persistent: one workgroup per CU (157.5 KB of LDS each)
    if (token_layout)
        hipLaunchKernelGGL((cost_patch_embed_kernel<H2, 80, true, F16, IN16, OUT16>), grid, dim3(256), P::LDS_BYTES, stream, cost_maps, (const char*)packed, out, S);
    else
        hipLaunchKernelGGL((cost_patch_embed_kernel<H2, 80, false, F16, IN16, OUT16>), grid, dim3(256), P::LDS_BYTES, stream, cost_maps, (const char*)packed, out, S);
    return mv_launch_status();
}

template <bool F16, bool IN16, bool OUT16>
static int launch_patch_embed_h(const void* cost_maps, const void* packed, void* out, int S, int H2, int token_layout, hipStream_t st) {
    return H2 == 60 ? launch_patch_embed<60, F16, IN16, OUT16>(cost_maps, packed, out, S, token_layout, st)
                    : launch_patch_embed<64, F16, IN16, OUT16>(cost_maps, packed, out, S, token_layout, st);
}

extern "C" int mv_cost_patch_embed_phase_ref(const void* cost_maps, int in_dtype, const void* packed, void* out, int out_dtype, int S, int H2, int W2,
                                             int token_layout, int operand_type, mvStream_t stream) {
    MV_CHECK_ARG(cost_maps && packed && out && S > 0 && (H2 == 60 || H2 == 64) && W2 == 80);
    MV_CHECK_ARG(operand_type == MV_F16 || operand_type == MV_BF16);
    MV_CHECK_ARG((in_dtype == MV_F32 || in_dtype == operand_type) && (out_dtype == MV_F32 || out_dtype == operand_type));
    if (in_dtype == MV_F32 && out_dtype != MV_F32) return MV_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const bool f16 = operand_type == MV_F16, in16 = in_dtype != MV_F32, out16 = out_dtype != MV_F32;
    if (f16) {
        if (!in16) return launch_patch_embed_h<true, false, false>(cost_maps, packed, out, S, H2, token_layout, st);
        return out16 ? launch_patch_embed_h<true, true, true>(cost_maps, packed, out, S, H2, token_layout, st)
                     : launch_patch_embed_h<true, true, false>(cost_maps, packed, out, S, H2, token_layout, st);
    }
    if (!in16) return launch_patch_embed_h<false, false, false>(cost_maps, packed, out, S, H2, token_layout, st);
    return out16 ? launch_patch_embed_h<false, true, true>(cost_maps, packed, out, S, H2, token_layout, st)
                 : launch_patch_embed_h<false, true, false>(cost_maps, packed, out, S, H2, token_layout, st);
}
#endif

#ifndef MV_PE_PHASE_REFERENCE
extern "C" int mv_cost_patch_embed(const float* cost_maps, const void* packed, float* out, int S, int H2, int W2, int token_layout,
                                   int operand_type, mvStream_t stream) {
    return mv_cost_patch_embed_t(cost_maps, MV_F32, packed, out, MV_F32, S, H2, W2, token_layout, operand_type, stream);
}
#endif
