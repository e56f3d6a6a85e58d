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

}  // namespace pe
