// Shared by patch_embed.hip (the two-slices-per-pass kernel for 60 / 64 x 80 slices) and patch_embed_v2.hip (the strip-mined kernel for any slice
// size): vector types, the packed-weight layout written by mv_patch_embed_pack, the 16-bit conversions and the MFMA wrapper.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace pe {

// packed weights (mv_patch_embed_pack): bf16 fragments in the B-operand order of v_mfma_f32_16x16x32 — lane l holds channel l % 16 of its 16-channel
// tile, k = 8 (l / 16) + 0..7 — then biases
//   [0, 3 KB)            conv1 (16x16x32 fragments: lane l = channel l % 16): 2 k-steps; k = (ky = 4 (g & 1) + (g >> 1) + 2 s, kx' = j), g = l / 16; zero for ky, kx' >= 6
//   [3 KB, 39 KB)        conv2: [channel tile 2][k-step 18]; k = (tap 2 ks + l / 32, cin 8 (l / 16 % 2) + j)
//   [39 KB, 183 KB)      conv3: [channel tile 4][tap 36]; k = cin 8 (l / 16) + j
//   then fp32 b1[32] (16 used), b2[32], b3[64]
constexpr size_t PE_W1_OFF = 0, PE_W2_OFF = 3 * 1024, PE_W3_OFF = 39 * 1024, PE_B_OFF = 183 * 1024, PE_PACKED_BYTES = PE_B_OFF + 128 * 4;

// the 16-bit operand type of the whole stack: bf16 (F16 = false) or IEEE fp16 (F16 = true; 11 significant bits — TF32's mantissa, and the type the
// reference's Fast mode runs this encoder in); same fragment layouts, same instruction shape
// IEEE half SATURATES at +-65504 (one v_med3_f32): a cost cell or an activation beyond fp16's range stays the largest finite value instead of
// becoming inf and, one layer later, NaN tokens (an un-normalised 256-channel dot product can get there; bf16 has fp32's range and needs nothing)
template <bool F16>
__device__ __forceinline__ uint16_t cvt_bits(float v) {
    if constexpr (F16) return __builtin_bit_cast(uint16_t, (_Float16)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f));
    else return __builtin_bit_cast(uint16_t, (__bf16)v);
}
template <bool F16>
__device__ __forceinline__ unsigned cvt_pack(float lo, float hi) { return (unsigned)cvt_bits<F16>(lo) | ((unsigned)cvt_bits<F16>(hi) << 16); }
template <bool F16>
__device__ __forceinline__ f32x4 mma16(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// the same product with the operands swapped: D^T = B^T A^T, i.e. a lane ends with FOUR CONSECUTIVE CHANNELS (rows 4 (l / 16) + e of the transposed tile) of ONE
// pixel (column l % 16) instead of four consecutive pixels of one channel — in the [8-channel chunk][16 B] cell layout of the LDS maps that is one 8-byte
// ds_write_b64 per tile instead of four ds_write_b16 (the A and B register layouts of v_mfma_f32_16x16x32 are the same function of the lane: row / column
// l % 16, k = 8 (l / 16) + j)
template <bool F16>
__device__ __forceinline__ f32x4 mma16t(bf16x8 act, bf16x8 wgt, f32x4 c) { return mma16<F16>(wgt, act, c); }

// LDS plan of the whole-slice kernels (patch_embed.hip: two slices per pass; patch_embed_v3.hip: producer / consumer wave groups) for 60 / 64 x 80 slices
template <int H2, int W2>
struct PE {
    static constexpr int HP = (H2 + 7) / 8 * 8, WP = (W2 + 7) / 8 * 8;
    static constexpr int H1 = HP / 2, W1 = WP / 2, H2o = HP / 4, W2o = WP / 4, H3 = HP / 8, W3 = WP / 8;
    static constexpr int M1 = H1 * W1, M2 = H2o * W2o, M3 = H3 * W3;
    static constexpr int T1 = M1 / 16, T2 = M2 / 16, T3 = M3 / 16;         // 16-pixel tiles (v_mfma_f32_16x16x32_bf16 everywhere)
    static_assert(M1 % 64 == 0 && T2 == 20 && T3 == 5, "five tiles per wave in conv2 (a quarter of the map) and conv3 (one slice)");
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
    // one (chunk, parity) plane.  A 16x16x32 fragment read has its four 16-lane groups in four DIFFERENT planes (lane / 16 = chunk / parity): the plane
    // stride decides whether they collide.  Simulated over every tile (4.0 LDS cycles per 64 lanes = conflict-free): conv1 map, pitch 26: no pad -> 4.0,
    // + 16 B -> 8.0; conv2 map, pitch 13: + 64 B -> 4.0, every other pad -> 8.0.
    static constexpr unsigned O1_PLANE = O1_ROWS * O1_XH * 16, O2_PLANE = O2_ROWS * O2_XH * 16 + 64;
    static_assert(O1_PLANE % 256 == 128 && O2_PLANE % 256 == 128, "plane strides as simulated (both land on half a bank row)");
    // byte offset of 8-channel chunk c of padded cell (row, col)
    static constexpr unsigned o1_cell(int c, int row, int col) { return (unsigned)(c * 2 + (col & 1)) * O1_PLANE + (unsigned)(row * O1_XH + (col >> 1)) * 16; }
    static constexpr unsigned o2_cell(int c, int row, int col) { return (unsigned)(c * 2 + (col & 1)) * O2_PLANE + (unsigned)(row * O2_XH + (col >> 1)) * 16; }
    static constexpr unsigned OFF_IN0 = 0, IN0_BYTES = IN_ROWS * IN_PITCH * 2;
    static constexpr unsigned OFF_O1 = OFF_IN0 + IN0_BYTES, O1_BYTES = 4 * O1_PLANE;      // 2 chunks x 2 parity planes
    static constexpr unsigned OFF_O2 = OFF_O1 + O1_BYTES, O2_BYTES = 8 * O2_PLANE;      // 4 chunks x 2 parity planes
    static constexpr unsigned OFF_W2B = OFF_O2 + 2 * O2_BYTES, W2B_BYTES = 18 * 1024;   // conv2 weight fragments of channel tile 1 (tile 0: registers)
    static constexpr unsigned LDS_BYTES = OFF_W2B + W2B_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS plan");
    static constexpr int Q4 = H2 * W2 / 4;                                 // float4s of a slice
    static_assert(W2 % 4 == 0 && Q4 <= 5 * 256, "slice staging: at most five float4 per thread");
};

}  // namespace pe
