// A13-A16 — MAC-VO 2D->3D covariance model, one 64-lane wave per keypoint
//
// Replaces Module/Covariance/Project2to3.py: MatchCovariance.estimate :124-181 (in-place clamp :131, patch gather
// :141-158, weighted mean/variance :161-172), Utility/Math.py:43-63 (gaussain_full_kernels), Covariance_2to3_full
// :377-423 (NED order z, x, y), create_3x3_matrix + .double() :426-433,:179, and optionally the world-frame
// rotation R cov R^T of Odometry/MACVO.py:273-281 (fp64 bmm(bmm(R, cov), R^T)).
//
// Reference quirk kept on purpose: local_filters[a][b] is the Gaussian at offset
// (x = a - h, y = b - h) with x <-> the FIRST covariance coordinate (u), while patches[a][b] is the depth at
// (v + a - h, u + b - h): the kernel is applied transposed relative to the patch.
//
// gfx950 design: latency-bound (0.8 MB for 200 keypoints).  A wave owns a keypoint: K*K taps are spread
// b-fastest over the lanes so each wave-wide load covers ~2 contiguous 124-B patch rows; taps and weights
// stay in registers (16 each for K = 31); three butterfly reductions (sum k, sum w*z, sum w*(z-mu)^2);
// lane 0 assembles the 3x3 in fp32 with the reference's op order, widens to fp64 and rotates.
#include "common.h"
#include <math.h>
#include "match_cov_dev.h"

using namespace mvcov;

namespace {

__global__ __launch_bounds__(256) void match_cov_kernel(CovSet s0, CovSet s1, mvMatchCovParams p, int cap, mvLaneCounts cnt) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int pl = blockIdx.z;
    if (n >= cnt.n[pl]) return;  // whole wave exits together
    match_cov_wave(blockIdx.y ? s1 : s0, p, cap, pl, n);   // the frame's two keypoint sets (kp0 on depth0, kp1 on depth1) share a launch
}

}  // namespace

extern "C" int mv_match_cov(const float* depth_map, const float* kp_uv, float* flow_cov, const float* depth_cov,
                            const double* rot, const mvMatchCovParams* params, int N, double* out_cov,
                            double* out_cov_rot, float* out_stats, mvStream_t stream) {
    MV_CHECK_ARG(params && N >= 0);
    if (N == 0) return MV_OK;
    MV_CHECK_ARG(depth_map && kp_uv && flow_cov && out_cov);
    const mvMatchCovParams p = *params;
    MV_CHECK_ARG(p.H > 0 && p.W > 0 && p.kernel_size >= 1 && (p.kernel_size & 1));
    if (p.kernel_size > MAX_K) return MV_ERR_UNSUPPORTED;
    MV_CHECK_ARG(p.use_patch_var || depth_cov);
    MV_CHECK_ARG(!out_cov_rot || rot);
    const CovSet s0{depth_map, kp_uv, flow_cov, depth_cov, rot, out_cov, out_cov_rot, out_stats};
    mvLaneCounts c{};
    c.n[0] = N;
    hipLaunchKernelGGL(match_cov_kernel, dim3(mv_ceil_div(N, 4), 1, 1), dim3(256), 0, (hipStream_t)stream, s0, s0, p, N, c);
    return mv_launch_status();
}

extern "C" int mv_match_cov_pair_lanes(const float* depth_map0, const float* kp_uv0, float* flow_cov0, const double* rot0,
                                       double* out_cov0, double* out_cov_rot0, const float* depth_map1,
                                       const float* kp_uv1, float* flow_cov1, double* out_cov1,
                                       const mvMatchCovParams* params, int lanes, const int32_t* n_live, int cap,
                                       mvStream_t stream) {
    MV_CHECK_ARG(params && lanes >= 1 && lanes <= MV_MAX_LANES && n_live && cap >= 0);
    mvLaneCounts c{};
    int n_max = 0;
    for (int l = 0; l < lanes; ++l) {
        MV_CHECK_ARG(n_live[l] >= 0 && n_live[l] <= cap);
        c.n[l] = n_live[l];
        n_max = n_live[l] > n_max ? n_live[l] : n_max;
    }
    if (n_max == 0) return MV_OK;
    MV_CHECK_ARG(depth_map0 && kp_uv0 && flow_cov0 && out_cov0 && depth_map1 && kp_uv1 && flow_cov1 && out_cov1);
    const mvMatchCovParams p = *params;
    MV_CHECK_ARG(p.H > 0 && p.W > 0 && p.kernel_size >= 1 && (p.kernel_size & 1) && p.use_patch_var);
    if (p.kernel_size > MAX_K) return MV_ERR_UNSUPPORTED;
    MV_CHECK_ARG(!out_cov_rot0 || rot0);
    const CovSet s0{depth_map0, kp_uv0, flow_cov0, nullptr, rot0, out_cov0, out_cov_rot0, nullptr};
    const CovSet s1{depth_map1, kp_uv1, flow_cov1, nullptr, nullptr, out_cov1, nullptr, nullptr};
    hipLaunchKernelGGL(match_cov_kernel, dim3(mv_ceil_div(n_max, 4), 2, lanes), dim3(256), 0, (hipStream_t)stream, s0, s1, p,
                       cap, c);
    return mv_launch_status();
}

extern "C" int mv_match_cov_pair(const float* depth_map0, const float* kp_uv0, float* flow_cov0, const double* rot0,
                                 double* out_cov0, double* out_cov_rot0, const float* depth_map1, const float* kp_uv1,
                                 float* flow_cov1, double* out_cov1, const mvMatchCovParams* params, int N,
                                 mvStream_t stream) {
    MV_CHECK_ARG(N >= 0);
    const int32_t n = N;
    return mv_match_cov_pair_lanes(depth_map0, kp_uv0, flow_cov0, rot0, out_cov0, out_cov_rot0, depth_map1, kp_uv1, flow_cov1,
                                   out_cov1, params, 1, &n, N, stream);
}
