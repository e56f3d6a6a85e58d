/*
 * macvo_hip.h — C ABI of libmacvo_hip.so: the MI355X (gfx950) hot path of MAC-VO.
 *
 * Every entry point is what a binding for the reference's per-frame hot path would call
 * (the reference is pure Python, so the "FFI" is ctypes — see INTEGRATION.md for the stub a
 * maintainer adds on the reference side).  Each function cites the reference code it replaces
 * (paths relative to the MAC-VO checkout).
 *
 * Conventions
 *   - All data pointers are DEVICE pointers unless the parameter says "host".
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls only enqueue
 *     work: no hidden synchronisation, no allocation, no host<->device copies.  The caller
 *     owns every buffer and synchronises when it needs results.
 *   - Return value: MV_OK (0) or a negative MV_ERR_* code; nothing throws.
 *   - Tensors are dense row-major with the shapes given; N1 = H1*W1, N2 = H2*W2.
 */
#ifndef MACVO_HIP_H
#define MACVO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mvStream_t;

enum {
    MV_OK = 0,
    MV_ERR_INVALID_ARG = -1,
    MV_ERR_UNSUPPORTED = -2,
    MV_ERR_LAUNCH = -3,
    MV_ERR_WORKSPACE = -4
};

enum {
    MV_F32 = 0,        /* fp32 operands, exact fp32 MFMA (bitwise an fmaf chain)                               */
    MV_F16 = 1,
    MV_BF16 = 2,
    MV_BF16X3 = 3,     /* fp32 operands pre-split by mv_split_bf16x3 into 3 bf16 planes; 6 bf16 MFMA products, fp32
                          accumulate: fp32-class accuracy (dropped terms O(2^-24)), not bitwise; layout HWC only     */
    MV_BF16X2 = 4,     /* the same planes, but only the two leading pieces (16 significant bits) and the 3 products
                          a0b1 + a1b0 + a0b0: relative error ~2^-16, finer than TF32 — the class the reference's fast
                          frontend runs this GEMM in (allow_tf32, Module/Frontend/Frontend.py:275-277); layout HWC only */
    MV_PACK_BF16X3 = 5,/* `mode` of mv_volume_pack / mv_corr_volume_packed: fp32 operands split into three bf16 pieces and
                          written in MFMA-fragment order; six piece products, fp32 accumulate (fp32-class, as MV_BF16X3) */
    MV_PACK_F16X2 = 6, /* the same with two fp16 pieces (11 + 11 bits) of every value after an exact power-of-two scaling of its
                          ROW (one pixel's feature vector) into fp16's range, three piece products, the scales undone exactly in
                          the epilogue: |error| <= ~2^-21 sum_k |a_k||b_k| — inside the exact path's parity bar, half the matrix
                          work of MV_PACK_BF16X3; the row scale is not clamped: any finite row of the fp32 range is carried */
    MV_VOL_ENC16 = 16  /* mvFramePipeConfig.volume_split only: 16-bit features, volume stored in the features' fp16 type */
};

/* feature-map memory layouts accepted by mv_corr_volume */
enum {
    MV_LAYOUT_CHW = 0, /* [B, C, N]  (NCHW feature maps, as the reference hands them over) */
    MV_LAYOUT_HWC = 1  /* [B, N, C]  (token-major / channels_last)                          */
};

/* most independent sequences ("lanes") one lane-batched launch / one frame pipe can carry (batch-32 frames = 32 lanes) */
#define MV_MAX_LANES 64

/* library ABI version (bumped on any signature change) and a static description string */
int mv_abi_version(void);
const char* mv_error_string(int code);

/* -------------------------------------------------------------------------------------------
 * A5  all-pairs cost volume.
 * Replaces FlowFormer `MemoryEncoder.corr` invoked at Module/Network/FlowFormerCov/flownet.py:26
 * (`corr = einsum('bhid,bhjd->bhij')`, heads = 1, no 1/sqrt(d)) and the `.float()` at flownet.py:27.
 *   out[b, i, j] = sum_c f1[b, c, i] * f2[b, c, j]          (== cost_maps [B*N1, 1, H2, W2] fp32)
 * in_dtype MV_F32: exact fp32 (v_mfma_f32_32x32x2_f32, one rounding per product, k ascending).
 * in_dtype MV_F16 / MV_BF16: 16-bit operands, fp32 accumulate, fp32 output (Fast mode).
 * in_dtype MV_BF16X3: f1/f2 are the [3][B][N][C] bf16 planes written by mv_split_bf16x3 from fp32 features
 *   (x = hi + mid + lo); the product is rebuilt from the 6 bf16 MFMA products with i + j <= 2.  The reference runs
 *   this GEMM in TF32/fp16 (Module/Frontend/Frontend.py:275-278), so this mode is at least as precise as its own.
 * Requirements: C % 16 == 0.  f1/f2 16-byte aligned.
 */
int mv_corr_volume(const void* f1, const void* f2, float* out, int B, int C, int N1, int N2,
                   int in_dtype, int layout, mvStream_t stream);

/* fp32 -> three bf16 planes (hi, mid, lo; residuals exact): planes[3][n] (uint16 bf16 bits), n % 4 == 0. */
int mv_split_bf16x3(const float* x, void* planes, size_t n, mvStream_t stream);
/* -------------------------------------------------------------------------------------------
 * (f)2 — the consumer of the cost volume: FlowFormer's cost PATCH EMBEDDING `proj` stack, fused (csrc/patch_embed.hip).
 * Replaces, per H2 x W2 slice of cost_maps (S = B*H1*W1 slices): F.pad to multiples of 8 -> Conv2d(1,16,6,stride 2,pad 2) -> ReLU ->
 * Conv2d(16,32,6,2,2) -> ReLU -> Conv2d(32,64,6,2,2) — PatchEmbed(patch_size 8, embed_dim 64) of FlowFormer's MemoryEncoder, reached from
 * Module/Network/FlowFormerCov/flownet.py:26; hyper-parameters Config/Train/Demo.yaml:20-36; token shape covhead.py:61-64.  The submodule's
 * source is absent from the reference checkout: shapes restated from the published FlowFormer sources (oracle/patch_embed.py), parity pinned
 * to F.conv2d.  16-bit matrix pipe, fp32 accumulate, intermediate maps in LDS (never in HBM).  operand_type = the 16-bit type of weights and
 * activations: MV_F16 (11 significant bits = TF32's mantissa, and the type MACVO_Fast.yaml:73 runs this encoder in) or MV_BF16 (bf16 encoders).
 *   mv_patch_embed_pack   OIHW fp32 weights + biases of the three Conv2d layers -> fragment-ordered 16-bit (+ fp32 biases), once per model
 *   mv_cost_patch_embed   cost_maps [S, H2, W2] fp32 -> token_layout ? [S, (H2/8)*(W2/8), 64] : [S, 64, H2/8, W2/8] fp32; operand_type must be
 *                         the one `packed` was built with
 *   mv_cost_patch_embed_supported   the slice sizes the LDS plan covers: 60 x 80 (640x480 frames) and 64 x 80 (the padded slice PatchEmbed.forward hands
 *                                   to `proj`; 640x512 frames); others: MV_ERR_UNSUPPORTED */
size_t mv_patch_embed_packed_bytes(void);
int mv_patch_embed_pack(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3, void* packed,
                        int operand_type, mvStream_t stream);
int mv_cost_patch_embed_supported(int H2, int W2);
int mv_cost_patch_embed(const float* cost_maps, const void* packed, float* out, int S, int H2, int W2, int token_layout, int operand_type,
                        mvStream_t stream);
/* the same with the slices and / or the tokens in the 16-bit OPERAND type (row (f)2 as SURVEY.md words it: "the volume kept in bf16 / fp16"): in_dtype =
 * MV_F32 or operand_type — fp16 cells of mv_corr_volume_out16's volume go to the matrix pipe as they are; out_dtype = MV_F32 or operand_type — tokens in the
 * encoder's dtype (what Conv2d under MACVO_Fast.yaml:73-74's fp16 autocast returns).  No widening pass on either side: 9.6 + 10.2 KB of HBM traffic per
 * 60 x 80 slice instead of 19.2 + 20.5 KB.  (in MV_F32, out 16-bit) is not built: MV_ERR_UNSUPPORTED.  fp16 conversions saturate at +-65504. */
int mv_cost_patch_embed_t(const void* cost_maps, int in_dtype, const void* packed, void* out, int out_dtype, int S, int H2, int W2, int token_layout,
                          int operand_type, mvStream_t stream);
/* -------------------------------------------------------------------------------------------
 * A5, split + streaming form (csrc/corr_volume_split.hip): the same volume from fp32 feature maps on the 16-bit matrix pipe.
 * gfx950 has no TF32 MFMA; the reference runs this GEMM in TF32 / fp16 (Module/Frontend/Frontend.py:275-277,
 * Config/Experiment/MACVO/MACVO_Fast.yaml:69-76).  mode MV_PACK_BF16X3: x = p0 + p1 + p2 (bf16 pieces, residuals exact), product
 * rebuilt from the six piece products with i + j <= 2 — same parity bar as the exact fp32 path, not bitwise.
 *
 * mv_volume_pack       both feature maps ([B,C,N] for MV_LAYOUT_CHW, [B,N,C] for MV_LAYOUT_HWC, fp32) -> packed operands
 *                      [B][ceil(N/32)+1 row blocks][C/16 k-steps][pieces][64 lanes][8 x 16 bit], one launch;
 *                      mv_volume_pack_bytes(B, C, N, mode) bytes each (0 = unsupported mode / shape).
 * mv_corr_volume_packed  out[b,i,j] fp32 [B,N1,N2] from two packed operands (C == 256, N2 % 64 == 0, N1*N2 < 2^30:
 *                      mv_corr_volume_packed_supported; MV_ERR_UNSUPPORTED otherwise — callers then use mv_corr_volume(MV_F32)).
 * Two entry points because the pack needs nothing but the feature maps: the frame driver issues it on another stream one
 * frame ahead of the GEMM. */
size_t mv_volume_pack_bytes(int B, int C, int N, int mode);
int mv_volume_pack(const float* f1, const float* f2, void* packed1, void* packed2, int B, int C, int N1, int N2, int layout,
                   int mode, mvStream_t stream);
/* mv_volume_pack with operand 2 (an H2 x W2 map, both multiples of 4) in 4 x 4-tile order: the GEMM's output columns then are that
 * order (see mv_corr_lookup_tiled); operand 1 and everything else as mv_volume_pack with N2 = H2 * W2 */
int mv_volume_pack_tiled(const float* f1, const float* f2, void* packed1, void* packed2, int B, int C, int N1, int H2, int W2,
                         int layout, int mode, mvStream_t stream);
int mv_corr_volume_packed_supported(int B, int C, int N1, int N2, int mode);
int mv_corr_volume_packed(const void* packed1, const void* packed2, float* out, int B, int C, int N1, int N2, int mode,
                          mvStream_t stream);
/* the same launch with `free_cus` compute units (rounded down to a multiple of 8) left without a persistent workgroup, for a caller that runs kernels
 * BESIDE the GEMM which cannot share a SIMD with its waves (mv_frame_pipe_*: the LM solve's workgroup).  Same bits for every value; 0 = mv_corr_volume_packed. */
int mv_corr_volume_packed_shared(const void* packed1, const void* packed2, float* out, int B, int C, int N1, int N2, int mode, int free_cus,
                                 mvStream_t stream);

/* Diagnostics: name of the kernel the calling thread's last mv_corr_volume dispatched ("" before the first call); the
 * string is static.  Used by the dispatch tests and by bench.py to name the kernel its roofline line is about. */
const char* mv_corr_volume_last_kernel(void);

/* -------------------------------------------------------------------------------------------
 * A6  (2r+1)^2 bilinear window lookup in each query's own cost slice.
 * Replaces FlowFormer `MemoryDecoder.encode_flow_token(cost_maps, coords1)` called at
 * Module/Network/FlowFormerCov/covhead.py:92 ("MUST run in fp32", :91) =
 * grid_sample(align_corners=True, padding_mode="zeros") on a [B*N1, 2r+1, 2r+1, 2] grid.
 *   vol    [B*N1, H2, W2] fp32;  coords [B, 2, H1, W1] fp32 (channel 0 = x, 1 = y)
 *   out    [B, (2r+1)^2, H1, W1] fp32; channel k = (2r+1)*i + j samples (x + i - r, y + j - r)
 * Requirements: 1 <= radius <= 4.
 */
int mv_corr_lookup(const float* vol, const float* coords, float* out, int B, int H1, int W1,
                   int H2, int W2, int radius, mvStream_t stream);

/* Fast mode (Config/Experiment/MACVO/MACVO_Fast.yaml:73-74: enc_dtype fp16): with 16-bit feature maps the reference's `einsum` returns the
 * volume in that 16-bit type and Module/Network/FlowFormerCov/flownet.py:27 merely widens it.  mv_corr_volume_out16 is that einsum with its ONE
 * rounding (fp32 accumulators -> round-to-nearest-even -> in_dtype) in the GEMM's epilogue and 2-byte cells: out [B, N1, N2] of in_dtype, half the
 * bytes of the output-bound kernel.  HWC feature maps, C = 128 / 256, N2 % 64 == 0 (mv_corr_volume_out16_supported; MV_ERR_UNSUPPORTED otherwise:
 * use mv_corr_volume + a cast).  mv_corr_lookup_vol16 is mv_corr_lookup on such an fp16 volume (cells widened on load = `cost_maps.float()`,
 * everything behind the load identical to the fp32 form); radius 4. */
int mv_corr_volume_out16_supported(int B, int C, int N1, int N2, int in_dtype, int layout);
int mv_corr_volume_out16(const void* f1, const void* f2, void* out, int B, int C, int N1, int N2, int in_dtype, int layout, mvStream_t stream);
int mv_corr_lookup_vol16(const void* vol_f16, const float* coords, float* out, int B, int H1, int W1, int H2, int W2, int radius,
                         mvStream_t stream);

/* The same lookup (same reference code, Module/Network/FlowFormerCov/covhead.py:92) on a volume whose per-query slices are stored in
 * 4 x 4-cell tiles: vol[(b N1 + q) H2 W2 + ((y / 4) * (W2 / 4) + x / 4) * 16 + (y % 4) * 4 + x % 4] — what mv_corr_volume_packed writes
 * when operand 2 was packed by mv_volume_pack_tiled.  Used inside the frame driver for >= 3 lanes, where it owns both the producer and
 * the consumer of the volume (the drop-in hook keeps the reference's [B*N1, 1, H2, W2]).  radius == 4, H2 % 4 == W2 % 4 == 0
 * (MV_ERR_UNSUPPORTED otherwise).  Results are bit-identical to mv_corr_lookup on the row-major volume. */
int mv_corr_lookup_tiled(const float* vol, const float* coords, float* out, int B, int H1, int W1,
                         int H2, int W2, int radius, mvStream_t stream);

/* Round 6: the tiled form for the Fast-mode volume's 2-byte cells.  mv_fmap_tile_rows16 writes the pixel rows of a 16-bit HWC feature map
 * [B, H, W, C] in 4 x 4-tile order (out[b][((y / 4) * (W / 4) + x / 4) * 16 + (y % 4) * 4 + x % 4][:] = f[b][y * W + x][:]; C % 8 == 0,
 * W % 4 == 0, out != f; ceil(H / 4) tile rows, the rows y >= H of the last one zero pixels: out is [B, mv_tiled_slice_cells(H, W), C]);
 * as operand 2 of mv_corr_volume_out16 (with N2 = mv_tiled_slice_cells) it makes the unchanged GEMM — the same `einsum` of
 * Module/Network/FlowFormerCov/flownet.py:26 — write every query's slice in the tile order above, where a tile is one aligned 32-byte
 * sector.  mv_corr_lookup_tiled_vol16 is mv_corr_lookup_vol16 (covhead.py:92 on `cost_maps.float()`) on that layout: ~338 B instead of
 * ~512 B per query, tokens bit-identical; a slice is mv_tiled_slice_cells(H2, W2) cells (the padding cells are exact zeros = the lookup's zero
 * padding).  vol 32-byte aligned; radius == 4, W2 % 4 == 0 (MV_ERR_UNSUPPORTED otherwise). */
int mv_tiled_slice_cells(int H, int W);   /* ceil(H / 4) * 4 * W; 0 when W % 4 != 0 */
int mv_fmap_tile_rows16(const void* f, void* out, int B, int C, int H, int W, mvStream_t stream);
int mv_corr_lookup_tiled_vol16(const void* vol_f16, const float* coords, float* out, int B, int H1, int W1, int H2, int W2, int radius,
                               mvStream_t stream);

/* -------------------------------------------------------------------------------------------
 * A8/A2  frontend epilogue: network output -> the typed records of IStereoDepth.Output / IMatcher.Output.
 * Replaces Module/Network/FlowFormerCov/flownet.py:44 (`exp(2*cov)`), Module/Frontend/Frontend.py:183-200
 * (`inference_2_depth`, `inference_2_match`), Module/Frontend/StereoDepth.py:270-282 and
 * Module/Frontend/Matching.py:28-40 (`from_partial_cov`).
 *   flow [2, 2, H, W] fp32 (sample 0 = stereo pair, sample 1 = temporal pair)
 *   logcov [2, 2, H, W] fp32 network log-sigma (cov = exp(2*logcov)); if cov_is_log == 0 it is already sigma^2
 *   outputs (all [H*W] planes, any may be NULL to skip):
 *   bl_fx = float(bl*fx), bl_fx_sq = float((bl*fx)**2), both evaluated in double by the caller exactly as
 *   the reference's Python scalars are, then rounded once to fp32 (what torch does with a Python scalar)
 *     disparity = |flow[0,0]|, disparity_cov = cov[0,0], depth = bl_fx * (1/disparity),
 *     depth_cov = bl_fx_sq * ((disparity_cov * (1/d^2)) / d^2), bad_mask = flow[0,0] <= 0 (uint8),
 *     match_flow [2,H,W] = flow[1], match_cov [3,H,W] = (cov[1,0], cov[1,1], 0)
 */
int mv_frontend_epilogue(const float* flow, const float* logcov, int cov_is_log, int H, int W,
                         float bl_fx, float bl_fx_sq, float* disparity, float* disparity_cov,
                         float* depth, float* depth_cov, uint8_t* bad_mask, float* match_flow,
                         float* match_cov, mvStream_t stream);

/* -------------------------------------------------------------------------------------------
 * §8(f) rank 1  convex 8x upsampling (RAFT / FlowFormer `upsample_flow`), optionally fused with exp(2*x).
 * Replaces MemoryDecoder.upsample_flow at Module/Network/FlowFormerCov/covhead.py:124-126,133-135 (in-tree twin
 * Module/Network/PWCNet/pwc_cov/gru.py:40-52) and, with exp2_out = 1, the exp(2*cov) of flownet.py:44.
 *   flow [B, 2, h, w] fp32 (1/8 resolution);  mask [B, 576, h, w] fp32 (channel = tap*64 + sy*8 + sx)
 *   out  [B, 2, 8h, 8w] fp32 = sum_tap softmax_tap(mask_scale * mask) * (8 * flow at the tap's 3x3 neighbour, zero pad)
 *   the softmax weights are exp(m - max) * (1 / sum): one division per sub-pixel
 *   mask_scale: 0.25 for the flow branch (covhead.py:121), 1.0 for the log-sigma branch (scaled inside CovUpdateBlock :41)
 */
int mv_convex_upsample(const float* flow, const float* mask, float* out, int B, int h, int w,
                       float mask_scale, int exp2_out, mvStream_t stream);
/* the same with the mask in the decoder's own type: mask_dtype MV_F32 / MV_F16 / MV_BF16 ([B, 576, h, w] of that type, read as it is and
 * widened in registers — the arithmetic is the fp32 formula on the widened values; Fast mode, Config/Experiment/MACVO/MACVO_Fast.yaml:73-74:
 * the mask head runs under autocast and no `.float()` copy of the 11 MB mask is made).  flow and out stay fp32. */
int mv_convex_upsample_m(const float* flow, const void* mask, int mask_dtype, float* out, int B, int h, int w,
                         float mask_scale, int exp2_out, mvStream_t stream);

/* -------------------------------------------------------------------------------------------
 * A10/A11 + MappingPointSelector  candidate generation for the covariance-aware keypoint selectors.
 * Replaces the dense part of Module/KeypointSelector.py: CovAwareSelector_NoDepth.select_point :362-400,
 * CovAwareSelector.select_point :260-327, MappingPointSelector.select_point :87-97 — quality map,
 * kernel_size x kernel_size min-NMS (NaN-propagating, equality test), border mask, (nan)median*1.5
 * thresholds with strict '<', optional validity masks, and torch.nonzero's row-major ordering.
 * The CPU `torch.randperm` (:331,:404,:98) stays on the host so indices are bit-exact; use
 * mv_kp_gather afterwards.
 */
enum { MV_KP_NODEPTH = 0, MV_KP_FULL = 1, MV_KP_MAPPING = 2 };

typedef struct {
    int32_t H, W;
    int32_t mode;          /* MV_KP_* */
    int32_t kernel_size;   /* odd, 1..15 (NMS window)            */
    int32_t mask_width;    /* border exclusion in pixels (>= 1)  */
    float max_depth;       /* FULL: z0 < max_depth & z1 < max_depth; MAPPING: z0 < max_depth */
    float max_depth_cov;   /* FULL: min(max_depth_cov, 1.5*nanmedian); MAPPING: plain '<'    */
    float max_match_cov;   /* NODEPTH/FULL: min(max_match_cov, 1.5*median)                   */
} mvKpSelectParams;

/* bytes of scratch mv_kp_select needs for an H x W image.  The workspace must be ZERO-FILLED once after allocation;
 * every call leaves its internal counters zeroed again, so it can be reused call after call (on one stream). */
size_t mv_kp_select_workspace_bytes(int H, int W);

/*
 *   flow_cov   [3, H, W] fp32 (uu, vv, uv) — required for NODEPTH, optional (may be NULL) for FULL
 *   depth0/1, depth0_cov/1_cov [H, W] fp32 — FULL needs all four, MAPPING needs depth0 + depth0_cov
 *   mask_a, mask_b [H, W] uint8 (0 = reject) or NULL — depth0_est.mask / match_est.mask
 *   out_cand   [H*W] int32: linear indices v*W + u of the surviving pixels in row-major order
 *   out_count  [4] int32: {n_candidates, n_nms, 0, 0}
 *   out_stats  [4] fp32: {median_flow_q, thresh_flow_q, median_depth0_cov, thresh_depth0_cov}
 */
int mv_kp_select(const float* flow_cov, const float* depth0, const float* depth0_cov,
                 const float* depth1, const float* depth1_cov, const uint8_t* mask_a,
                 const uint8_t* mask_b, const mvKpSelectParams* params /* host */, void* workspace,
                 size_t workspace_bytes, int32_t* out_cand, int32_t* out_count, float* out_stats,
                 mvStream_t stream);

/* selected[perm][..., 2:].roll(1, 1)  (KeypointSelector.py:331-332,404-405):
 *   out_uv[k] = (cand[perm[k]] % W, cand[perm[k]] / W) as int64 (u, v) */
int mv_kp_gather(const int32_t* cand, const int64_t* perm, int n_sel, int W, int64_t* out_uv,
                 mvStream_t stream);

/* -------------------------------------------------------------------------------------------
 * A12  keypoint tracking + the scalar-map gathers of Odometry/MACVO.py:198-232 in one launch.
 *   kp1 = kp0 + flow[:, v0, u0]; inbound = edge < u1 < W - edge & edge < v1 < H - edge (strict,
 *   Utility/Point.py:5-13); gathers at kp0 (integer) and at kp1 truncated toward zero
 *   (Module/Frontend/Frontend.py:117); match covariance read at the SOURCE pixel kp0 (:231);
 *   kp0 gets the constant sigma (match_cov_default, match_cov_default, 0) (:228-229).
 * Outputs are written for every input keypoint (row order preserved); `inbound` says which rows the
 * reference keeps.  kp0_uv int64 [N,2].  Planes are [H*W] fp32; match_cov is [3,H*W].
 *   out_kp0 [N,2] fp32 (= kp0 as float, or NULL); out_kp1 [N,2] fp32; out_inbound [N] uint8;
 *   out_vals [11, N] fp32 (SoA rows) = d0, disp0, sdisp0, sdd0, d1, disp1, sdisp1, sdd1, suu, svv, suv
 *   (for rows that are not inbound the kp1-side values are 0; absent maps give -1 like the reference)
 *   out_sigma0, out_sigma1 [N,3] fp32 or NULL: (uu, vv, uv) of kp0 / of the match, ready for mv_match_cov
 */
int mv_kp_track(const int64_t* kp0_uv, int N, const float* match_flow, const float* match_cov,
                const float* depth0, const float* disp0, const float* sdisp0, const float* sdd0,
                const float* depth1, const float* disp1, const float* sdisp1, const float* sdd1,
                int H, int W, int edge, float match_cov_default, float* out_kp0, float* out_kp1,
                uint8_t* out_inbound, float* out_vals, float* out_sigma0, float* out_sigma1,
                mvStream_t stream);

/* pixel2point_NED + world transform of Odometry/MACVO.py:240,273-281 (Utility/Point.py:15-17, pp.pixel2point):
 *   pos_Tc[n] = (d, ((u-cx)*d)/fx, ((v-cy)*d)/fy)   (fp32, NED)          kp_uv [N,2] fp32
 *   pos_Tw[n] = pose.Act(pos_Tc[n])                  (fp32, PyPose SO3_Act + t)  pose [7] fp32 device
 *   rot [9] fp64 = pose.rotation().matrix().to(float64) (matrix evaluated in fp32 like PyPose, then widened)
 * depth_vals[n * depth_stride] lets the caller pass a column of the mv_kp_track table directly.
 * Any of pos_Tc / pos_Tw / rot may be NULL. */
int mv_backproject(const float* kp_uv, const float* depth_vals, int depth_stride, float fx, float fy,
                   float cx, float cy, const float* pose, int N, float* pos_Tc, float* pos_Tw,
                   double* rot, mvStream_t stream);

/* Observation filters of Module/OutlierFilter.py fused into one validity mask (row order kept, no compaction):
 *   flags bit0 CovarianceSanityFilter :91-100 (no NaN/Inf in obs1_covTc / obs2_covTc)
 *         bit1 SimpleDepthFilter :103-121 (min_depth <= d1, d2 <= max_depth)
 *         bit2 LikelyFrontOfCamFilter :124-137 (d - 2*sqrt(sigma_d) > 0 for both observations)
 *   inbound [N] uint8 or NULL (border test of mv_kp_track); vals = the [11, N] SoA table of mv_kp_track
 *   valid [N] uint8; count [1] int32 = number of valid rows */
int mv_obs_filter(const uint8_t* inbound, const double* cov1, const double* cov2, const float* vals,
                  int flags, float min_depth, float max_depth, int N, uint8_t* valid, int32_t* count,
                  mvStream_t stream);

/* -------------------------------------------------------------------------------------------
 * A13-A16  MAC-VO covariance model.
 * Replaces Module/Covariance/Project2to3.py: MatchCovariance.estimate :124-181 (incl. the in-place
 * clamp of flow_cov :131), Utility/Math.py:43-63 (gaussain_full_kernels), Covariance_2to3_full
 * :377-423, create_3x3_matrix :426-433, and (optionally) the world-frame rotation
 * cov_Tw = R cov R^T of Odometry/MACVO.py:273-281.
 */
typedef struct {
    int32_t H, W;
    int32_t kernel_size;       /* odd, <= 31 */
    int32_t use_patch_var;     /* 1: weighted patch variance (flow_cov given or depth_cov NULL); 0: use depth_cov */
    float fx, fy, cx, cy;
    float min_flow_cov_sq;     /* config.min_flow_cov ** 2 (clamp applied to flow_cov[:, :2] IN PLACE) */
    float min_depth_cov;
} mvMatchCovParams;

/*
 *   depth_map [H, W] fp32; kp_uv [N, 2] fp32 (u, v) — patch centre = trunc(kp), (u - cx) uses the float
 *   flow_cov [N, 3] fp32 in/out (clamped in place); depth_cov [N] fp32 or NULL
 *   rot [9] fp64 row-major or NULL
 *   out_cov [N, 9] fp64 NED order (z, x, y); out_cov_rot [N, 9] fp64 = R cov R^T or NULL
 *   out_stats [N, 2] fp32 = (weighted mean depth, clamped variance) or NULL
 */
int mv_match_cov(const float* depth_map, const float* kp_uv, float* flow_cov,
                 const float* depth_cov, const double* rot, const mvMatchCovParams* params /* host */,
                 int N, double* out_cov, double* out_cov_rot, float* out_stats, mvStream_t stream);

/* Dense-mapping tail of run_pair (Odometry/MACVO.py:313-337, `mapping: true`): for the map pixels chosen by
 * MappingPointSelector (mv_kp_select MV_KP_MAPPING + mv_kp_gather, 2000 per frame) gather depth (:317) and its variance
 * (:320), pixel2point_NED (:318), prev_pose.Act (:334), the constant match sigma (:322-323) and optionally the colour
 * (imageL*255 -> uint8, :326-328).  The covariance of these points is one mv_match_cov call on out_uv / out_sigma
 * (:324 — note the reference stores it UNROTATED, in the camera frame).
 *   uv [N,2] int64; depth, depth_cov [H*W] fp32; image [3,H,W] fp32 in [0,1] or NULL; pose fp32[7] (device)
 *   out_uv [N,2], out_d [N], out_sdd [N], out_sigma [N,3], out_Tc [N,3], out_Tw [N,3] fp32, out_color [N,3] uint8 (NULL ok) */
int mv_map_points(const int64_t* uv, int N, const float* depth, const float* depth_cov, const float* image, int H, int W,
                  float fx, float fy, float cx, float cy, const float* pose, float match_cov_default, float* out_uv,
                  float* out_d, float* out_sdd, float* out_sigma, float* out_Tc, float* out_Tw, uint8_t* out_color,
                  mvStream_t stream);

/* Both ObsCovModel.estimate calls of one frame (Odometry/MACVO.py:241-242) in ONE launch: set 0 = kp0 on the previous
 * frame's depth map (+ optional world rotation, :273-281), set 1 = kp1 on the current depth map.  Same arithmetic as
 * two mv_match_cov calls with use_patch_var = 1 (both flow_cov arrays are clamped in place). */
int mv_match_cov_pair(const float* depth_map0, const float* kp_uv0, float* flow_cov0, const double* rot0,
                      double* out_cov0, double* out_cov_rot0, const float* depth_map1, const float* kp_uv1,
                      float* flow_cov1, double* out_cov1, const mvMatchCovParams* params /* host */, int N,
                      mvStream_t stream);

/* -------------------------------------------------------------------------------------------
 * A17-A22  covariance-weighted two-frame pose-graph solve, batched over independent problems.
 * Replaces TwoFrame_PGO._optimize (Module/Optimization/TwoFramePGO/Optimizer.py:81-102), the residual
 * graphs + analytic Jacobians (Module/Optimization/TwoFramePGO/Graphs.py:33-231), LM_analytic.step
 * (Module/Optimization/PyposeOptimizers.py:160-194) and the PyPose 0.6.8 pieces they call
 * (Huber, FastTriggs, TrustRegion, PINV, StopOnPlateau, SE3 Exp/Act/Inv).  All arithmetic in fp64
 * after the fp32 buffer construction the reference performs (Optimizer.py:84-85).
 */
enum { MV_GRAPH_ICP = 0, MV_GRAPH_REPROJ = 1, MV_GRAPH_DISP = 2 };

typedef struct {
    double huber_delta;  /* 0.1   */
    double radius;       /* 1e3   (damping0 = 1/radius) */
    double tr_high, tr_low, tr_up, tr_down, tr_factor, tr_min, tr_max; /* .5 1e-3 2 .5 .5 1e-6 1e16 */
    double diag_min, diag_max; /* 1e-6 1e32 */
    double decreasing;   /* 1e-5  */
    double pinv_rcond;   /* 1e-15 */
    int32_t reject;      /* 16    */
    int32_t max_steps;   /* 10    */
    int32_t patience;    /* 2     */
    int32_t stop_on_reject; /* 1: StopOnPlateau ends the outer loop when the last step's reject_count >= this value
                             * (PyPose 0.6.x scheduler: `if optimizer.reject_count > 0` => 1, the default; `reject` (16)
                             * = "only after the maximum number of rejections"; 0 = rejections never stop the loop).
                             * PyPose is un-vendored (requirements.txt:1) => from memory, hence a named knob. */
} mvLMParams;

void mv_lm_default_params(mvLMParams* p /* host */);

/*
 * Problem p owns points [offsets[p], offsets[p+1]) of the concatenated per-point arrays.
 *   offsets [nprob+1] int32; init_pose [nprob,7] fp32 (tx ty tz qx qy qz qw);
 *   intrinsics [nprob,4] fp32 (fx fy cx cy); baseline [nprob] fp32
 *   pos_Tw [Ntot,3] fp32; cov_Tw [Ntot,9] fp64 (ICP only, else may be NULL)
 *   pixel2_uv [Ntot,2] fp32; pixel2_d [Ntot] fp32 (ICP); pixel2_disp, pixel2_disp_cov [Ntot] fp32 (DISP)
 *   pixel2_uv_cov [Ntot,3] fp32 (REPROJ/DISP); obs2_covTc [Ntot,9] fp64 (ICP)
 *   valid [Ntot] uint8 or NULL: rows with 0 are skipped (the reference drops them from the map instead:
 *     Odometry/MACVO.py:200-206 border filter, :269-270 outlier filter); min_points: a problem with fewer valid
 *     rows is not optimised and returns init_pose with steps = 0 (lost track, MACVO.py:303-307; pass 0 to disable)
 *   out_pose [nprob,7] fp64; out_info [nprob,4] fp64 = {final loss, outer steps, last reject_count, initial loss}
 *   out_pose_f32 [nprob,7] fp32 or NULL: `motion.float()` of write_graph_data (Optimizer.py:104-108)
 */
int mv_pgo_solve(int nprob, const int32_t* offsets, int graph_type, const float* init_pose,
                 const float* intrinsics, const float* baseline, const float* pos_Tw,
                 const double* cov_Tw, const float* pixel2_uv, const float* pixel2_d,
                 const float* pixel2_disp, const float* pixel2_disp_cov, const float* pixel2_uv_cov,
                 const double* obs2_covTc, const uint8_t* valid, int min_points,
                 const mvLMParams* params /* host */, double* out_pose, double* out_info,
                 float* out_pose_f32, mvStream_t stream);
/* mv_pgo_solve with the rest of the backend folded into the same launch (Module/OutlierFilter.py:91-141, Odometry/MACVO.py:273-281,
 * Optimizer.py:81-102).  Before a problem's rows are read its workgroup
 *   (1) filter_flags >= 0: runs the observation filters of mv_obs_filter_lanes over the problem's rows (inbound, cov_Tc = cov1, obs2_covTc = cov2, the
 *       [11, nprob, cap] value table `vals`; cap = rows per problem) and writes `valid_out` / `count_out` [nprob] — the solve then reads valid_out;
 *       filter_flags < 0: `valid` is an input as in mv_pgo_solve;
 *   (2) rotates the first n_live[prob] rows into the world frame with the problem's init_pose — pos_Tw = T p_cam (fp32 PyPose SE3 Act), cov_Tw = R cov_Tc
 *       R^T (fp64), out_rot [nprob, 9] = R — mv_pose_apply_lanes' arithmetic, written to pos_Tw / cov_Tw (cov_Tc / cov_Tw may both be NULL).
 * Same bits as the separate kernels.  pose_sink (or NULL): a second fp32 copy of the optimised poses [nprob, 7].  nprob <= MV_MAX_LANES.  The filters address a problem's
 * rows through `offsets` like the rest of the launch (rows [offsets[l], offsets[l + 1]), the value table's row stride = offsets[nprob]); the frame driver passes
 * the static table {0, cap, 2 cap, ...}.  n_live: host int32 [nprob], n_live[l] <= offsets[l + 1] - offsets[l]. */
int mv_pgo_solve_posed(int nprob, const int32_t* offsets, const int32_t* n_live, int cap, int graph_type, const float* init_pose,
                       const float* intrinsics, const float* baseline, const float* pos_Tc, const double* cov_Tc,
                       float* pos_Tw, double* cov_Tw, double* out_rot, const float* pixel2_uv, const float* pixel2_d,
                       const float* pixel2_disp, const float* pixel2_disp_cov, const float* pixel2_uv_cov,
                       const double* obs2_covTc, int filter_flags, float filter_min_depth, float filter_max_depth,
                       const uint8_t* inbound, const float* vals, uint8_t* valid, int32_t* count_out, int min_points,
                       const mvLMParams* params, double* out_pose, double* out_info, float* out_pose_f32, float* pose_sink,
                       mvStream_t stream);
/* ... of the device-driven frame (round 6): the live-row count of problem l is n_live_dev[l * n_live_stride] in DEVICE memory (<= cap; written by
 * mv_backend_front_draw_lanes earlier on the stream) — the number of selected keypoints never reaches the host.  Everything else as above. */
int mv_pgo_solve_posed_dev(int nprob, const int32_t* offsets, const int32_t* n_live_dev, int n_live_stride, int cap, int graph_type,
                           const float* init_pose, const float* intrinsics, const float* baseline, const float* pos_Tc, const double* cov_Tc,
                           float* pos_Tw, double* cov_Tw, double* out_rot, const float* pixel2_uv, const float* pixel2_d,
                           const float* pixel2_disp, const float* pixel2_disp_cov, const float* pixel2_uv_cov, const double* obs2_covTc,
                           int filter_flags, float filter_min_depth, float filter_max_depth, const uint8_t* inbound, const float* vals,
                           uint8_t* valid, int32_t* count_out, int min_points, const mvLMParams* params, double* out_pose, double* out_info,
                           float* out_pose_f32, float* pose_sink, mvStream_t stream);

/* -------------------------------------------------------------------------------------------
 * A23  PWC-Net local correlation, forward (the reference's only hand-written CUDA kernel; non-default matcher path).
 * Replaces `FunctionCorrelation(tenFirst, tenSecond)` = `_FunctionCorrelation.forward`
 * (Module/Network/PWCNet/pwc/correlation.py:277-325; kernels :8-33 rearrange, :35-103 updateOutput):
 *   out[b, 9*(dy+4)+(dx+4), y, x] = (1/C) * sum_c first[b,c,y,x] * second[b,c,y+dy,x+dx],  dx,dy in [-4,4], zero padded.
 * first, second: [B, C, H, W] fp32 contiguous (NCHW, as the reference asserts :283-284); out: [B, 81, H, W] fp32.
 * Callers: pwc_model.py:178-233 (five pyramid levels), pwc_model_tartanvo.py:231,259, RAFTCov.py:7.
 */
int mv_local_corr81(const float* first, const float* second, float* out, int B, int C, int H, int W,
                    mvStream_t stream);

/* -------------------------------------------------------------------------------------------
 * SURVEY §8(f) rank 4  device-resident VisualMap (tracking map of one sequence), output poses, MotionInterpolate.
 * Replaces Module/Map/VisualMap.py:15-133 + Module/Map/Graph.py:19-298 as driven by Odometry/MACVO.py:158-171,244-311
 * (MatchObs.init, `match_obs[mask]`, points.push(...[mask]), push_keyframe, the six edge updates, the lost-track flag), which
 * the reference runs on the CPU behind ~25 `.cpu()` copies per frame; Odometry/Interface.py:47-49 (body poses of poses.npy);
 * Module/MapProcessor.py:52-76 (MotionInterpolate) with Utility/Math.py:96-133.  The stores are caller-owned device arrays
 * (SoA, the reference's field names / dtypes, VisualMap.py:23-69); `counts` is a DEVICE int64[6] = {frames, matches, points,
 * lost frames, refused appends, map points} advanced by the kernel itself, so registering a frame needs no host synchronisation: the host only has to keep
 * capacity >= an upper bound of the rows pushed (rows selected).
 */
typedef struct {
    /* frames  [cap_f, .] */
    float* K;                 /* [.,3,3] */
    float* baseline;          /* [.]     */
    float* pose;              /* [.,7]   sensor-to-world; prior at push time, overwritten by the optimised pose */
    float* T_BS;              /* [.,7]   */
    uint8_t* need_interp;     /* [.]     */
    int64_t* time_ns;         /* [.]     */
    /* points  [cap_p, .] */
    float* pos_Tw;            /* [.,3]   */
    double* cov_Tw;           /* [.,3,3] */
    uint8_t* color;           /* [.,3]   */
    /* match   [cap_m, .]  (VisualMap.py:51-69) */
    float *pixel1_uv, *pixel2_uv;                    /* [.,2] */
    float *pixel1_d, *pixel2_d, *pixel1_disp, *pixel2_disp, *pixel1_disp_cov, *pixel2_disp_cov;   /* [.,1] */
    double *obs1_covTc, *obs2_covTc;                 /* [.,3,3] */
    float *pixel1_uv_cov, *pixel2_uv_cov;            /* [.,3] */
    float *pixel1_d_cov, *pixel2_d_cov;              /* [.,1] */
    /* edges (Graph.py: DenseEdge_Multi ranges [n, max_deg, 2] + num [n]; SingleEdge mapping [n]; SparseEdge_Multi edges
     * [n, max_deg] + deg [n]); all int64, -1 = empty */
    int64_t *frame2match_ranges, *frame2match_num, *frame2map_ranges, *frame2map_num;
    int64_t *match2frame1, *match2frame2, *match2point;
    int64_t *point2match_edges, *point2match_deg;
    int64_t* counts;          /* device int64[6]: {frames, matches, points, lost frames, REFUSED appends (error word), map points} */
    int32_t max_pt_obs;       /* 5  (VisualMap.py:18) */
    int32_t max_frame_range;  /* 2  (VisualMap.py:19) */
    /* capacities (rows) of the frame / match / point stores and their edge tables.  The row offsets come from the DEVICE-side
     * counts, so the kernel itself refuses an append that would not fit (counts unchanged, counts[4] += 1, out_frame_idx = -1)
     * instead of writing past the stores; a frame2match range dropped because a frame already has max_frame_range ranges (the
     * reference raises there, Graph.py:183-186) is counted in counts[4] as well.  The host reads counts[4] at its next
     * synchronisation point (DeviceVisualMap.sizes -> MV_ERR_WORKSPACE). */
    int64_t cap_frames, cap_match, cap_points;
    /* dense map points of `mapping: true` (VisualMap.map_points, VisualMap.py:47-54; indexed by the frame2map edges) */
    float* mp_pos_Tw;         /* [cap_mp,3]   */
    double* mp_cov_Tw;        /* [cap_mp,3,3] camera-frame covariance, stored unrotated as the reference does (MACVO.py:324,334) */
    uint8_t* mp_color;        /* [cap_mp,3]   */
    int64_t cap_map_points;   /* 0 = no map-point store attached */
} mvMapStores;

/* one frame's observations as the tracking kernels leave them (row order of the selected keypoints; `valid` = border test
 * AND outlier filter, i.e. the rows the reference keeps, MACVO.py:200-206,269-270) */
typedef struct {
    int32_t n_rows;           /* selected keypoints of the frame (0 for the very first frame)           */
    int32_t table_stride;     /* row stride of the SoA `vals` table (>= n_rows)                          */
    int32_t prev_frame;       /* map index of the previous keyframe, -1 for the first frame (initialize) */
    int32_t min_num_point;    /* fewer kept rows => need_interp (MACVO.py:303-307)                       */
    const uint8_t* valid;     /* [n_rows] or NULL (keep all)                                             */
    const float *kp0, *kp1;   /* [n_rows,2] pixel1_uv / pixel2_uv                                        */
    const float* vals;        /* [11, table_stride] table of mv_kp_track                                 */
    const float *sigma0, *sigma1;       /* [n_rows,3] pixel1_uv_cov / pixel2_uv_cov (after the in-place clamp)  */
    const double *cov0, *cov1;          /* [n_rows,9] obs1_covTc / obs2_covTc                                   */
    const float* pos_Tw;      /* [n_rows,3]  */
    const double* cov0_world; /* [n_rows,9] R cov0 R^T (MACVO.py:273-281)                                */
    const uint8_t* color;     /* [n_rows,3] or NULL                                                      */
    const float* K;           /* [9] device */
    const float* T_BS;        /* [7] device */
    const float* prior_pose;  /* [7] device or NULL (identity): the pose the frame is pushed with         */
    float baseline;
    int64_t time_ns;
    int32_t* out_frame_idx;   /* device int32[1] or NULL: the map index the frame received                */
} mvMapFrame;

int mv_map_append(const mvMapFrame* frame /* host */, const mvMapStores* stores /* host */, mvStream_t stream);
/* Dense-mapping tail (Odometry/MACVO.py:329-337): `map_points.push(PointNode.init({pos_Tw, cov_Tw, color}))` +
 * `frame2map.add(frame_idx, num_map_orig, n)` for the NEWEST registered frame (counts[0] - 1).  n rows of pos_Tw [n,3] fp32,
 * cov [n,9] fp64, color [n,3] uint8 (NULL = zeros), all device.  Refused (counts[4] += 1) when the store or the frame's range
 * slots are full. */
int mv_map_append_points(const mvMapStores* stores /* host */, int n, const float* pos_Tw, const double* cov, const uint8_t* color,
                         mvStream_t stream);
/* out[i] = T_BS[i] @ pose[i] @ T_BS[i]^-1 in fp32 (Odometry/Interface.py:47-49) */
int mv_body_poses(const float* pose, const float* T_BS, int T, float* out, mvStream_t stream);
/* MotionInterpolate.elaborate_map on pose [T,7] in place (fp64 inside); scratch: double[7*(T-1)]; out_count int32[1] or NULL =
 * number of interpolated motions */
int mv_motion_interpolate(float* pose, const uint8_t* need_interp, int T, double* scratch, int32_t* out_count,
                          mvStream_t stream);

/* -------------------------------------------------------------------------------------------
 * Lane-batched variants: the same kernels over `lanes` INDEPENDENT frames (sequences) in ONE launch.  This is the
 * reference's batching point (Module/Frontend/Frontend.py:219-224 concatenates pairs along the batch axis) carried through
 * the rest of run_pair, for BASELINE configs[4] (batch-32 frames per GPU).  Layout rule: every argument gains a leading
 * [lanes] dimension (maps [lanes, ch, H, W]; per-keypoint tables [lanes, cap, .] with `cap` rows of capacity per lane of
 * which n_live[l] are live — n_live is a HOST int32[lanes], travels as a kernel argument), EXCEPT the SoA value table of
 * mv_kp_track, which is [11, lanes, cap] so that each of its rows is one concatenated per-point column that mv_pgo_solve
 * takes directly (problem l = rows [l*cap, (l+1)*cap), dead rows masked by `valid`, which mv_obs_filter_lanes zeroes).
 * The plain entry points above are these with lanes = 1, cap = N.  MV_KP_MAPPING supports lanes = 1 only. */
int mv_frontend_epilogue_lanes(const float* flow, const float* logcov, int cov_is_log, int H, int W, float bl_fx,
                               float bl_fx_sq, float* disparity, float* disparity_cov, float* depth, float* depth_cov,
                               uint8_t* bad_mask, float* match_flow, float* match_cov, int lanes, mvStream_t stream);
/* workspace: lanes consecutive copies of mv_kp_select_workspace_bytes(H, W), zero-filled once */
int mv_kp_select_lanes(const float* flow_cov, const float* depth0, const float* depth0_cov, const float* depth1,
                       const float* depth1_cov, const uint8_t* mask_a, const uint8_t* mask_b,
                       const mvKpSelectParams* params /* host */, void* workspace, size_t workspace_bytes,
                       int32_t* out_cand /* [lanes, H*W] */, int32_t* out_count /* [lanes, 4] */,
                       float* out_stats /* [lanes, 4] */, int lanes, mvStream_t stream);
/* The pose-dependent remainder of a frame's backend (Odometry/MACVO.py:273-281), split off so that gather / track / back-projection /
 * covariances / filters can run before the previous frame's solve has finished: pos_Tw = T * pos_Tc (fp32 SE3 Act),
 * rot [lanes, 9] = R as fp64, cov_rot = R cov R^T (fp64).  Bit-identical to what mv_backproject(pose) and mv_match_cov(rot,
 * out_cov_rot) produce.  pose [lanes, 7] device; tables [lanes, cap, .]; pos_* / cov* pairs may be null together. */
int mv_pose_apply_lanes(const float* pose, const float* pos_Tc, const double* cov, int lanes, const int32_t* n_live /* host */,
                        int cap, float* pos_Tw, double* rot, double* cov_rot, mvStream_t stream);
/* mv_frontend_epilogue_lanes + mv_kp_select_lanes (MV_KP_NODEPTH) in one launch less: the selector's first kernel computes the
 * quality sigma_uu + sigma_vv from the network's covariance planes itself and writes the epilogue's maps for its own pixels.
 * Same results as the two calls in sequence, bit for bit (Frontend.py:183-200 + KeypointSelector.py:362-407).  Other
 * selector modes read the previous frame's maps too: MV_ERR_UNSUPPORTED. */
int mv_frontend_epilogue_select_lanes(const float* flow, const float* logcov, int cov_is_log, float bl_fx, float bl_fx_sq,
                                      float* disparity, float* disparity_cov, float* depth, float* depth_cov,
                                      uint8_t* bad_mask, float* match_flow, float* match_cov, const uint8_t* mask_a,
                                      const uint8_t* mask_b, const mvKpSelectParams* params /* host */, void* workspace,
                                      size_t workspace_bytes, int32_t* out_cand, int32_t* out_count, float* out_stats,
                                      int lanes, mvStream_t stream);
int mv_kp_gather_lanes(const int32_t* cand, size_t cand_lane_stride, const int64_t* perm /* [lanes, cap] */, int lanes,
                       const int32_t* n_live /* host */, int cap, int W, int64_t* out_uv /* [lanes, cap, 2] */,
                       mvStream_t stream);
/* mv_kp_gather_lanes + mv_kp_track_lanes + mv_backproject_lanes (camera frame, depth = the depth0 value the tracking gather
 * read) of a frame in ONE launch: same expressions, same bits (tests/test_gpu_backend.py::test_kp_front_equals_the_three_calls).
 * perm: device [lanes, cap] or — one lane, <= 256 rows — host: the indices then travel inside the kernel arguments (no copy). */
int mv_kp_front_lanes(const int32_t* cand, size_t cand_lane_stride, const int64_t* perm_dev, const int64_t* perm_host /* or NULL */,
                      int lanes, const int32_t* n_live /* host */, int cap, const float* match_flow, const float* match_cov,
                      const float* depth0, const float* disp0, const float* sdisp0, const float* sdd0, const float* depth1,
                      const float* disp1, const float* sdisp1, const float* sdd1, int H, int W, int edge, float match_cov_default,
                      float fx, float fy, float cx, float cy, int64_t* out_kp0_uv /* [lanes, cap, 2] */, float* out_kp0,
                      float* out_kp1, uint8_t* out_inbound, float* out_vals /* [11, lanes, cap] */, float* out_sigma0,
                      float* out_sigma1, float* out_pos_Tc /* [lanes, cap, 3] */, mvStream_t stream);
/* The pose-INDEPENDENT half of a frame's backend in one launch: mv_kp_front_lanes + mv_match_cov_pair_lanes (patch variance, no
 * rotation) — Odometry/MACVO.py:197-262, Module/Covariance/Project2to3.py:124-181 — with the same results bit for bit.  A wave = one keypoint of one of
 * the two keypoint sets.  cov_params: H, W, intrinsics, kernel size and clamps (use_patch_var must be 1).  The observation filters that follow are the
 * prologue of mv_pgo_solve_posed. */
int mv_backend_front_lanes(const int32_t* cand, size_t cand_lane_stride, const int64_t* perm_dev, const int64_t* perm_host, int lanes,
                           const int32_t* n_live, int cap, const float* match_flow, const float* match_cov, const float* depth0,
                           const float* disp0, const float* sdisp0, const float* sdd0, const float* depth1, const float* disp1,
                           const float* sdisp1, const float* sdd1, int edge, float match_cov_default, const mvMatchCovParams* cov_params,
                           int64_t* out_kp0_uv, float* out_kp0, float* out_kp1, uint8_t* out_inbound, float* out_vals, float* out_sigma0,
                           float* out_sigma1, float* out_pos_Tc, double* out_cov0, double* out_cov1, mvStream_t stream);
/* mv_backend_front_lanes of the device-driven frame (round 6): `selected_points[torch.randperm(n)[:numPoint]]` (Module/KeypointSelector.py:331,404) without
 * the host.  n = count_dev[l * count_stride] is read from device memory (where mv_kp_select_lanes left it: count_stride = 4), the permutation head is drawn
 * inside the launch from lane l's device-resident MT19937 state_in[l] (mv_mt19937_seed; same bits as torch's CPU generator + torch.randperm) and the advanced
 * generator is written to state_out[l] (a DIFFERENT buffer: every workgroup reads state_in).  out_perm [lanes, cap] int64 = the head, out_live [lanes, 2] =
 * {min(n, num_point), n}; rows beyond min(n, num_point) of the tables stay untouched.  num_point <= mv_randperm_max_head(). */
int mv_backend_front_draw_lanes(const int32_t* cand, size_t cand_lane_stride, const int32_t* count_dev, int count_stride, const uint32_t* state_in,
                                uint32_t* state_out, int num_point, int lanes, int cap, const float* match_flow, const float* match_cov,
                                const float* depth0, const float* disp0, const float* sdisp0, const float* sdd0, const float* depth1,
                                const float* disp1, const float* sdisp1, const float* sdd1, int edge, float match_cov_default,
                                const mvMatchCovParams* cov_params, int64_t* out_perm, int32_t* out_live, int64_t* out_kp0_uv, float* out_kp0,
                                float* out_kp1, uint8_t* out_inbound, float* out_vals, float* out_sigma0, float* out_sigma1, float* out_pos_Tc,
                                double* out_cov0, double* out_cov1, mvStream_t stream);
int mv_kp_track_lanes(const int64_t* kp0_uv, int lanes, const int32_t* n_live /* host */, int cap, const float* match_flow,
                      const float* match_cov, const float* depth0, const float* disp0, const float* sdisp0,
                      const float* sdd0, const float* depth1, const float* disp1, const float* sdisp1, const float* sdd1,
                      int H, int W, int edge, float match_cov_default, float* out_kp0, float* out_kp1,
                      uint8_t* out_inbound, float* out_vals /* [11, lanes, cap] */, float* out_sigma0, float* out_sigma1,
                      mvStream_t stream);
/* depth of lane l, row n = depth_vals[l * depth_lane_stride + n * depth_stride]; pose [lanes, 7]; rot [lanes, 9] */
int mv_backproject_lanes(const float* kp_uv, const float* depth_vals, int depth_stride, size_t depth_lane_stride, float fx,
                         float fy, float cx, float cy, const float* pose, int lanes, const int32_t* n_live /* host */,
                         int cap, float* pos_Tc, float* pos_Tw, double* rot, mvStream_t stream);
int mv_match_cov_pair_lanes(const float* depth_map0, const float* kp_uv0, float* flow_cov0, const double* rot0,
                            double* out_cov0, double* out_cov_rot0, const float* depth_map1, const float* kp_uv1,
                            float* flow_cov1, double* out_cov1, const mvMatchCovParams* params /* host */, int lanes,
                            const int32_t* n_live /* host */, int cap, mvStream_t stream);
/* vals = the [11, lanes, cap] table; valid [lanes, cap] (rows >= n_live[l] are written 0); count [lanes] */
int mv_obs_filter_lanes(const uint8_t* inbound, const double* cov1, const double* cov2, const float* vals, int flags,
                        float min_depth, float max_depth, int lanes, const int32_t* n_live /* host */, int cap,
                        uint8_t* valid, int32_t* count, mvStream_t stream);

/* -------------------------------------------------------------------------------------------
 * Native per-frame driver: the host-side sequencing of one `MACVO.run_pair` (Odometry/MACVO.py:173-311) and
 * `FlowFormerCovFrontend.estimate_pair` (Module/Frontend/Frontend.py:215-232) for the hot path, as two host calls per
 * frame instead of ~30 (a Python loop over the entry points above is interpreter-bound at ~370 us/frame, more than the
 * GPU work).  Internally: 4 HIP streams (volume GEMM | lookups+epilogue+selector | tracking..filter | LM solve),
 * rotating buffer slots carved out of ONE caller-provided device arena, events for every cross-stream hazard.
 *
 * Call order (frame 0 = MACVO.initialize :158-171, depth only):
 *     mv_frame_pipe_enqueue(p, &in0, s, 0);
 *     mv_frame_pipe_enqueue(p, &in1, s, 1);
 *     loop t = 1..: mv_frame_pipe_enqueue(p, &in[t+1], s, 1);          // next frame's frontend first (software pipeline)
 *                   mv_frame_pipe_enqueue_volume(p, &in[t+2], s);        // (optional) and the GEMM of the one after
 *                   mv_frame_pipe_wait_candidates(p, &n);                // host blocks on frame t's selector only
 *                   perm = torch.randperm(n)[:num_point]                 // stays on the host CPU: bit-exact indices
 *                   mv_frame_pipe_finish(p, perm, n_sel, pose_sink);     // backend + solve of frame t
 * At most three tracked frames may be in flight (enqueued, not finished): candidates x3, maps x5, volumes x3 rotate.
 *
 * Lanes (BASELINE configs[4], "batch-32 frames per GPU"): with pairs = 2 * lanes the pipe advances `lanes` INDEPENDENT
 * sequences in lock-step through the same launches — one volume GEMM of 2 * lanes pairs (pair 2l = lane l's stereo pair,
 * 2l + 1 its temporal pair, the batch axis of Frontend.py:219-220), lane-batched lookups / epilogue / selector / backend
 * kernels, and ONE mv_pgo_solve over `lanes` problems.  Every input of mvFrameInputs and every reported buffer then has
 * the leading dimension scaled by `lanes`; wait_candidates / finish / set_pose take per-lane arrays.
 */
typedef struct mvFramePipe mvFramePipe;

typedef struct {
    int32_t H, W;              /* image size, multiples of 8 */
    int32_t C;                 /* feature channels (% 16 == 0) */
    int32_t pairs;             /* volume batch = 2 * lanes (<= 2 * MV_MAX_LANES): pair 2l stereo, 2l + 1 temporal of lane l */
    int32_t iters;             /* decoder iterations = window lookups per frame (12) */
    int32_t radius;            /* lookup radius (4) */
    int32_t feat_dtype;        /* MV_F32 | MV_F16 | MV_BF16 */
    int32_t layout;            /* MV_LAYOUT_* of the feature maps */
    int32_t volume_split;      /* MV_VOL_ENC16 (16): fp16 features, volume STORED in fp16 as the reference's Fast mode computes it
                                  (mv_corr_volume_out16 + mv_corr_lookup_vol16; shapes outside that kernel's domain keep fp32 cells) |
                                  0 exact fp32 | MV_PACK_F16X2 (6, the host side's default) | MV_PACK_BF16X3 (5): fp32 features of either
                                  layout packed on the device (mv_volume_pack, beside the previous frame's GEMM) +
                                  mv_corr_volume_packed, shapes it does not cover run the exact kernel | 3 | 2: fp32 HWC features through the round-1 plane split, multiplied
                                  as MV_BF16X3 / MV_BF16X2 */
    int32_t selector_mode;     /* MV_KP_NODEPTH | MV_KP_FULL */
    int32_t kp_kernel_size, kp_mask_width;
    int32_t num_point;         /* keypoints per frame (200) */
    int32_t edgewidth;         /* strict border of the tracked keypoints (32) */
    int32_t min_num_point;     /* fewer valid observations -> pose not optimised (MACVO.py:64,303-307) */
    int32_t graph_type;        /* MV_GRAPH_* */
    int32_t filters;           /* mv_obs_filter flags */
    int32_t cov_kernel_size;   /* 31 */
    int32_t mapping;           /* 1: dense-mapping tail of run_pair (Odometry/MACVO.py:313-337; `mapping: true`), lanes == 1 */
    int32_t map_num_point;     /* 2000 (:315) */
    int32_t map_mask_width;    /* MappingPointSelector args (Config/Experiment/MACVO/MACVO_Fast.yaml) */
    int32_t async_backend;     /* backend launch thread: the ~10 launches of `finish` (+ the permutation draw of finish_seeded) are issued by
                                  a second host thread while the caller's thread enqueues the next frame.  1 on, -1 off, 0 = environment
                                  MV_PIPE_ASYNC_BACKEND if set, else the library default.  Results are identical either way. */
    float fx, fy, cx, cy, baseline;
    float bl_fx, bl_fx_sq;     /* baseline*fx and its square, rounded once from double (StereoDepth.py:270-282) */
    float match_cov_default, max_match_cov, max_depth_cov, max_depth;
    float min_flow_cov_sq, min_depth_cov, filter_min_depth;
    float map_max_depth, map_max_depth_cov;   /* MappingPointSelector: z < map_max_depth, sigma_z^2 < map_max_depth_cov (KeypointSelector.py:87-100) */
    mvLMParams lm;
} mvFramePipeConfig;

/* what the learned layers hand over for one estimate_pair (device pointers, fp32 unless noted) */
typedef struct {
    const void* fmap1;      /* [pairs, C, H/8, W/8] (CHW) or [pairs, H/8, W/8, C] (HWC), feat_dtype */
    const void* fmap2;
    const float* coords;    /* [iters, pairs, 2, H/8, W/8]: coords1 entering each decoder iteration (covhead.py:85-92) */
    const float* flow;      /* [pairs, 2, H, W] last upsampled flow        } either these two ...            */
    const float* logcov;    /* [pairs, 2, H, W] last upsampled log-sigma   }                                  */
    const float* flow8;     /* [pairs, 2, H/8, W/8]   } ... or the 1/8-resolution fields + convex-upsampling */
    const float* cov8;      /* [pairs, 2, H/8, W/8]   }     masks of the last iteration (covhead.py:119-135); */
    const float* up_mask;   /* [pairs, 576, H/8, W/8] }     up_mask BEFORE its 0.25 scale, cov_mask after      */
    const float* cov_mask;  /* [pairs, 576, H/8, W/8] }                                                       */
} mvFrameInputs;

/* buffers reported by mv_frame_pipe_buffer (element counts, not bytes; MV_FB_VOLUME's element is the volume CELL: fp32, or fp16 — 2 bytes —
 * when the pipe was created with volume_split = MV_VOL_ENC16); each has a leading [lanes] dimension, per-keypoint
 * tables are [lanes, num_point, .] (a lane's live rows = its n_sel), MV_FB_VALS is [11, lanes, num_point] */
enum {
    MV_FB_VOLUME = 0, MV_FB_TOKENS, MV_FB_DISPARITY, MV_FB_DISPARITY_COV, MV_FB_DEPTH, MV_FB_DEPTH_COV, MV_FB_MATCH_FLOW,
    MV_FB_MATCH_COV, MV_FB_CAND, MV_FB_COUNT, MV_FB_STATS,                       /* frontend side: age counts enqueued frames */
    MV_FB_KP0, MV_FB_KP0F, MV_FB_KP1, MV_FB_INBOUND, MV_FB_VALS, MV_FB_SIGMA0, MV_FB_SIGMA1, MV_FB_POS_TC, MV_FB_POS_TW,
    MV_FB_ROT, MV_FB_COV0, MV_FB_COV0W, MV_FB_COV1, MV_FB_VALID, MV_FB_NVALID, MV_FB_POSE64, MV_FB_INFO,   /* backend side */
    MV_FB_POSE,                                                                  /* fp32 [lanes, 7]; age 0 = newest solve's output */
    /* dense-mapping tail of the newest finished frame (mapping = 1; rows = what mv_frame_pipe_map_points was called with) */
    MV_FB_MAP_UV, MV_FB_MAP_D, MV_FB_MAP_SDD, MV_FB_MAP_TC, MV_FB_MAP_TW, MV_FB_MAP_COV /* fp64 [.,9] */, MV_FB_MAP_COLOR /* u8 [.,3] */,
    /* device-driven frame (round 6): the permutation head the front launch drew, int64 [lanes, num_point]; int32 [lanes, 2] = selected keypoints, candidates */
    MV_FB_PERM, MV_FB_LIVE
};

size_t mv_frame_pipe_arena_bytes(const mvFramePipeConfig* cfg);           /* 0 = invalid configuration */
int mv_frame_pipe_max_pending(void);   /* how many tracked frames may be enqueued and not yet finished (the slot rotation this library was built with) */
/* how many tracked frames the host should keep in flight for this pipe shape (the stream layout the driver will pick for `lanes` sequences with / without
 * the dense-mapping tail): 3 for the two-decoder-stream layout of one- and two-lane pipes and for batched pipes, 2 for the classic one-lane layout */
int mv_frame_pipe_default_depth(int lanes, int mapping);
/* arena: device memory, 256-byte aligned, >= mv_frame_pipe_arena_bytes; must outlive the pipe */
int mv_frame_pipe_create(const mvFramePipeConfig* cfg, void* arena, size_t arena_bytes, mvFramePipe** out);
void mv_frame_pipe_destroy(mvFramePipe* p);
int mv_frame_pipe_set_pose(mvFramePipe* p, const float* pose7_host);      /* [lanes, 7] host: priors of the next frame (blocking) */
/* frontend half of a frame; inputs must be complete on `in_stream` (an event is recorded there) and stay untouched
 * until the frame's lookups ran.  with_selector = 0 for the very first frame. */
int mv_frame_pipe_enqueue(mvFramePipe* p, const mvFrameInputs* in, mvStream_t in_stream, int with_selector);
/* optional: issue the volume GEMM of the NEXT frame (the one the next mv_frame_pipe_enqueue will complete) right away; it
 * only needs fmap1 / fmap2 and a free volume buffer, so the host can queue it before it blocks on the previous frame's
 * candidate count and the GEMM stream never waits for the host.  At most one GEMM ahead. */
int mv_frame_pipe_enqueue_volume(mvFramePipe* p, const mvFrameInputs* in, mvStream_t in_stream);
int mv_frame_pipe_wait_candidates(mvFramePipe* p, int32_t* n_cand /* [lanes] host */);   /* oldest unfinished frame; blocks the host */
/* perm_host: int64 [lanes, num_point], row l = randperm(n_cand[l])[:num_point] (n_sel[l] entries used); n_sel: int32 [lanes]
 * host; pose_sink: device fp32 [lanes, 7] or NULL (copy of the new poses) */
int mv_frame_pipe_finish(mvFramePipe* p, const int64_t* perm_host, const int32_t* n_sel, float* pose_sink);
/* Native keypoint permutations.  The reference draws `torch.randperm(n)[:numPoint]` from torch's CPU generator
 * (Module/KeypointSelector.py:331,404) = MT19937 + Fisher-Yates (ATen randperm_cpu).  mv_frame_pipe_seed_lanes gives every lane
 * its own MT19937 seeded like `torch.Generator().manual_seed(seed)`; mv_frame_pipe_finish_seeded then replaces
 * wait_candidates + host randperm + finish by one call (first numPoint swaps, the remaining draws discarded: same bits, ~20x
 * less host time — what a 32-lane step needs).  n_cand_out / n_sel_out: [lanes] host or NULL. */
/* Dense-mapping tail (config.mapping = 1, lanes == 1; Odometry/MACVO.py:303-337).  The reference maps only when tracking
 * succeeded and then draws its SECOND randperm of the frame, so the host has to see the frame's observation count first:
 *   mv_frame_pipe_wait_tracked   blocks until the newest finished frame's filter count is on the host; returns it and the number
 *                                of MappingPointSelector candidates (found on the PREVIOUS frame's depth maps, on the decoder-side
 *                                stream next to the tracking selector);
 *   mv_frame_pipe_map_points     perm_host = randperm(n_cand_map)[:map_num_point]: gather, depth / variance gathers,
 *                                pixel2point_NED, prev_pose.Act, constant match sigma, colours (image_dev [3,H,W] fp32 in [0,1] or
 *                                NULL), the 31x31 covariance model (stored unrotated, :324) — into the MV_FB_MAP_* buffers — and,
 *                                with `stores`, map_points.push + frame2map.add (mv_map_append_points) behind this frame's
 *                                mv_frame_pipe_map_append.  Call order per frame: finish, [map_append], wait_tracked, [map_points]. */
int mv_frame_pipe_wait_tracked(mvFramePipe* p, int32_t* n_valid, int32_t* n_cand_map);
int mv_frame_pipe_map_points(mvFramePipe* p, const int64_t* perm_host, int n_sel, const float* image_dev,
                             const mvMapStores* stores /* host, or NULL */);
int mv_frame_pipe_seed_lanes(mvFramePipe* p, const uint64_t* seeds /* [lanes] host */);
/* host-only (no GPU): `calls` successive torch.randperm(n[i], generator=g)[:k] of ONE generator g = torch.Generator().manual_seed(seed),
 * as the seeded finish draws them for a lane (Module/KeypointSelector.py:331,404); out [calls, k] (row i: min(k, n[i]) entries) */
int mv_randperm_heads(uint64_t seed, const int64_t* n, int calls, int k, int64_t* out);
int mv_frame_pipe_finish_seeded(mvFramePipe* p, float* pose_sink, int32_t* n_cand_out, int32_t* n_sel_out);
/* Device-resident permutations (round 6).  The same draw — torch.randperm(n)[:k] of an MT19937 seeded like torch.Generator().manual_seed(seed),
 * Module/KeypointSelector.py:331,404 — made ON THE GPU from a candidate count that never leaves it: no D2H count, no host generator, no H2D permutation.
 * A lane's generator lives in device memory as mv_randperm_state_words() uint32 words (the 624-word block + the position of the next draw);
 *   mv_mt19937_seed            fills that representation on the host (copy it to the device);
 *   mv_randperm_head_lanes     one workgroup per lane: n = n_dev[l * n_stride]; writes out_perm[l * cap + (0 .. min(n, k)))  and
 *                              out_n_sel[l * n_sel_stride] = min(n, k); advances the lane's generator by max(n - 1, 0) draws (k <= mv_randperm_max_head());
 *   mv_randperm_heads_emulated host-only (no GPU): the same phase functions (csrc/randperm_dev.h) run thread by thread with `threads` emulated
 *                              threads — `calls` successive draws of one generator, out [calls, k] like mv_randperm_heads. */
int mv_randperm_state_words(void);
int mv_randperm_max_head(void);
int mv_mt19937_seed(uint64_t seed, uint32_t* state_host);
int mv_randperm_head_lanes(uint32_t* state, const int32_t* n_dev, int n_stride, int lanes, int k, int cap, int64_t* out_perm, int32_t* out_n_sel,
                           int n_sel_stride, mvStream_t stream);
int mv_randperm_heads_emulated(uint64_t seed, const int64_t* n, int calls, int k, int threads, int64_t* out);
/* The device-driven frame.  mv_frame_pipe_seed_lanes also places the generators in device memory and — unless MV_PIPE_DEVICE_DRAW=0, the dense-mapping tail is on or
 * num_point > mv_randperm_max_head() — switches the pipe to it (mv_frame_pipe_device_draw() == 1): a frame is then finished with mv_frame_pipe_finish_device,
 * which never waits: backend + solve are queued behind the frame's selector by event and the front launch (mv_backend_front_draw_lanes) draws the permutation
 * itself.  Same keypoints, same poses as mv_frame_pipe_finish_seeded.  mv_frame_pipe_finished_counts: the counts of the age-th newest finished frame
 * (0 or 1), for whoever needs them on the host (blocks until that frame's front launch has run). */
int mv_frame_pipe_device_draw(const mvFramePipe* p);
/* 1 when the pipe's volume buffers (the VOLUME view) hold every query's slice in 4 x 4-cell tiles (mv_corr_lookup_tiled / _tiled_vol16 read them):
 * Fast-mode pipes by default, split-precision pipes under MV_PIPE_TILED=1 */
int mv_frame_pipe_volume_tiled(const mvFramePipe* p);
int mv_frame_pipe_host_threads(const mvFramePipe* p);   /* 1: the caller issues everything; 2: + the backend launch thread (default where the process has >= 3 cores) */
int mv_frame_pipe_finish_device(mvFramePipe* p, float* pose_sink);
int mv_frame_pipe_finished_counts(mvFramePipe* p, int age, int32_t* n_cand, int32_t* n_sel);
/* host-side flow control of a device-driven stream: blocks until the front launch of the finish `lag` finishes back (0 = the newest; lag <= 6) has run */
int mv_frame_pipe_wait_finished(mvFramePipe* p, int lag);
/* register the newest FINISHED frame in a device-resident map (mv_map_append on the pipe's own streams, no copies; lanes = 1):
 * frame_idx = the map index the frame receives (= frames pushed so far), prev_frame = the previous keyframe's index; the
 * optimised pose is written over the frame's prior once its solve has finished */
int mv_frame_pipe_map_append(mvFramePipe* p, const mvMapStores* stores /* host */, int frame_idx, int prev_frame,
                             const float* K_dev, const float* T_BS_dev, float baseline, int64_t time_ns,
                             const uint8_t* color_dev /* [n_sel,3] or NULL */);
/* Result views and slot reuse: a consumer that reads result views (mv_frame_pipe_buffer) asynchronously on its own stream
 * calls this before it finishes the next frame; the pipe then orders the kernels that recycle those buffers behind
 * everything enqueued on `stream` so far.  (Host-synchronous consumers do not need it.) */
int mv_frame_pipe_release(mvFramePipe* p, mvStream_t stream);
/* block_host = 1: wait for all four streams on the host; 0: make `stream` wait for everything enqueued so far (including the
 * frontends of frames enqueued ahead); 2: make `stream` wait for the newest FINISHED frame's backend + solve only — what a
 * consumer of that frame's results needs while later frames are already queued */
int mv_frame_pipe_sync(mvFramePipe* p, mvStream_t stream, int block_host);
/* measurement hook (bench.py roofline): record a HIP-event pair around each of the next max_launches volume GEMMs on the
 * stream they run on (0 = off; restarts the count), and read the elapsed milliseconds back (blocks on that stream) */
int mv_frame_pipe_time_volume(mvFramePipe* p, int max_launches);
int mv_frame_pipe_volume_times(mvFramePipe* p, float* ms, int cap, int* n);
/* start of each timed GEMM, ms since the first timed one (same events) */
int mv_frame_pipe_volume_starts(mvFramePipe* p, float* ms, int cap, int* n);
/* on = 0: a timed frame records ONLY the event pair around its volume GEMM (what the roofline needs), not the six timeline events on the other three
 * streams — each is a barrier packet on a stream whose launch chain bounds a one-lane pipe; mv_frame_pipe_timeline[_backend] then report no frames.
 * Default 1.  bench.py times the contract's K steps with 0 and collects the timeline in a separate untimed pass. */
int mv_frame_pipe_time_detail(mvFramePipe* p, int on);
/* per timed frame: {GEMM start, GEMM end, last lookup done, selector done} in ms since the first timed GEMM start */
int mv_frame_pipe_timeline(mvFramePipe* p, float* ms, int cap_frames, int* n);
/* ... and {backend start, backend end, pose_apply start, solve end} of the same frames (-1: not finished / no keypoints) */
int mv_frame_pipe_timeline_backend(mvFramePipe* p, float* ms, int cap_frames, int* n);
/* where a result lives inside the arena; age 0 = newest frame that passed that stage, 1 = the one before */
int mv_frame_pipe_buffer(mvFramePipe* p, int which, int age, void** ptr, size_t* count);

#ifdef __cplusplus
}
#endif
#endif /* MACVO_HIP_H */
