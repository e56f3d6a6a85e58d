// A8 / A2 / A12 — frontend epilogue and keypoint tracking gathers
//
// mv_frontend_epilogue replaces Module/Network/FlowFormerCov/flownet.py:44 (exp(2*cov)),
//   Module/Frontend/Frontend.py:183-200 (inference_2_depth / inference_2_match),
//   Module/Frontend/StereoDepth.py:270-282 (disparity_to_depth, disparity_to_depth_cov) and
//   Module/Frontend/Matching.py:28-40 (from_partial_cov) with ONE pass over the network output
//   (the reference runs ~8 elementwise torch kernels, two of them torch.compile'd).
// mv_kp_track replaces Odometry/MACVO.py:198-232 (kp1 = kp0 + flow[kp0], strict border filter of
//   Utility/Point.py:5-13, and the ten retrieve_pixels gathers of Module/Frontend/Frontend.py:103-118).
#include "common.h"
#include <math.h>

namespace {

__global__ __launch_bounds__(256) void frontend_epilogue_kernel(
    const float* __restrict__ flow, const float* __restrict__ logcov, int cov_is_log, int plane, float bl_fx,
    float bl_fx_sq, float* __restrict__ disparity, float* __restrict__ disparity_cov, float* __restrict__ depth,
    float* __restrict__ depth_cov, uint8_t* __restrict__ bad_mask, float* __restrict__ match_flow,
    float* __restrict__ match_cov) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += gridDim.x * blockDim.x) {
        // sample 0 (stereo pair): flow[0,0] -> disparity, cov[0,0] -> disparity variance
        const float fx0 = flow[i];
        const float lc0 = logcov[i];
        const float dcov = cov_is_log ? expf(lc0 * 2.f) : lc0;
        const float d = fabsf(fx0);
        if (disparity) disparity[i] = d;
        if (disparity_cov) disparity_cov[i] = dcov;
        if (depth) depth[i] = bl_fx * (1.f / d);
        if (depth_cov) {
            const float d2 = d * d;
            const float err2 = dcov * (1.f / d2);
            depth_cov[i] = bl_fx_sq * (err2 / d2);
        }
        if (bad_mask) bad_mask[i] = fx0 <= 0.f;
        // sample 1 (temporal pair): flow[1] and cov[1] padded with sigma_uv = 0
        if (match_flow) {
            match_flow[i] = flow[2 * plane + i];
            match_flow[plane + i] = flow[3 * plane + i];
        }
        if (match_cov) {
            const float l0 = logcov[2 * plane + i], l1 = logcov[3 * plane + i];
            match_cov[i] = cov_is_log ? expf(l0 * 2.f) : l0;
            match_cov[plane + i] = cov_is_log ? expf(l1 * 2.f) : l1;
            match_cov[2 * plane + i] = 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void kp_track_kernel(const int64_t* __restrict__ kp0_uv, int N,
                                                       const float* __restrict__ match_flow,
                                                       const float* __restrict__ match_cov,
                                                       const float* __restrict__ depth0, const float* __restrict__ disp0,
                                                       const float* __restrict__ sdisp0, const float* __restrict__ sdd0,
                                                       const float* __restrict__ depth1, const float* __restrict__ disp1,
                                                       const float* __restrict__ sdisp1, const float* __restrict__ sdd1,
                                                       int H, int W, int edge, float* __restrict__ out_kp1,
                                                       uint8_t* __restrict__ out_inbound, float* __restrict__ out_vals) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int plane = H * W;
    const int u0 = (int)kp0_uv[2 * n], v0 = (int)kp0_uv[2 * n + 1];
    const bool ok0 = u0 >= 0 && u0 < W && v0 >= 0 && v0 < H;
    const int i0 = ok0 ? v0 * W + u0 : 0;
    // int64 + float32 -> float32 (torch type promotion)
    const float u1 = (float)u0 + match_flow[i0];
    const float v1 = (float)v0 + match_flow[plane + i0];
    const bool inb = ok0 && (u1 < (float)(W - edge)) && (u1 > (float)edge) && (v1 < (float)(H - edge)) && (v1 > (float)edge);
    out_kp1[2 * n] = u1;
    out_kp1[2 * n + 1] = v1;
    out_inbound[n] = inb;
    float* o = out_vals + (size_t)n * 11;
    o[0] = depth0[i0];
    o[1] = disp0 ? disp0[i0] : -1.f;
    o[2] = sdisp0 ? sdisp0[i0] : -1.f;
    o[3] = sdd0 ? sdd0[i0] : -1.f;
    if (inb) {
        const int i1 = (int)v1 * W + (int)u1;  // .long(): truncation toward zero
        o[4] = depth1[i1];
        o[5] = disp1 ? disp1[i1] : -1.f;
        o[6] = sdisp1 ? sdisp1[i1] : -1.f;
        o[7] = sdd1 ? sdd1[i1] : -1.f;
    } else {
        o[4] = o[5] = o[6] = o[7] = 0.f;
    }
    // match covariance is read at the SOURCE pixel kp0 (MACVO.py:231)
    o[8] = match_cov ? match_cov[i0] : -1.f;
    o[9] = match_cov ? match_cov[plane + i0] : -1.f;
    o[10] = match_cov ? match_cov[2 * plane + i0] : -1.f;
}

}  // namespace

extern "C" int mv_frontend_epilogue(const float* flow, const float* logcov, int cov_is_log, int H, int W,
                                    float bl_fx, float bl_fx_sq, float* disparity, float* disparity_cov, float* depth,
                                    float* depth_cov, uint8_t* bad_mask, float* match_flow, float* match_cov,
                                    mvStream_t stream) {
    MV_CHECK_ARG(flow && logcov && H > 0 && W > 0);
    const int plane = H * W;
    const int blocks = min(mv_ceil_div(plane, 256), 2048);
    hipLaunchKernelGGL(frontend_epilogue_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, flow, logcov,
                       cov_is_log, plane, bl_fx, bl_fx_sq, disparity, disparity_cov, depth, depth_cov, bad_mask,
                       match_flow, match_cov);
    return mv_launch_status();
}

extern "C" int mv_kp_track(const int64_t* kp0_uv, int N, const float* match_flow, const float* match_cov,
                           const float* depth0, const float* disp0, const float* sdisp0, const float* sdd0,
                           const float* depth1, const float* disp1, const float* sdisp1, const float* sdd1, int H,
                           int W, int edge, float* out_kp1, uint8_t* out_inbound, float* out_vals,
                           mvStream_t stream) {
    MV_CHECK_ARG(N >= 0 && H > 0 && W > 0 && edge >= 0);
    if (N == 0) return MV_OK;
    MV_CHECK_ARG(kp0_uv && match_flow && depth0 && depth1 && out_kp1 && out_inbound && out_vals);
    hipLaunchKernelGGL(kp_track_kernel, dim3(mv_ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, kp0_uv, N,
                       match_flow, match_cov, depth0, disp0, sdisp0, sdd0, depth1, disp1, sdisp1, sdd1, H, W, edge,
                       out_kp1, out_inbound, out_vals);
    return mv_launch_status();
}
