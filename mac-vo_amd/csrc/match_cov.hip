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

namespace {

constexpr int MAX_K = 31;
constexpr int MAX_TAPS_PER_LANE = (MAX_K * MAX_K + 63) / 64;  // 16

__device__ __forceinline__ float clamp_min_nanprop(float x, float m) { return (x < m) ? m : x; }  // torch.clamp(min=)

struct CovSet {
    const float* depth_map;
    const float* kp_uv;
    float* flow_cov;
    const float* depth_cov;
    const double* rot;
    double* out_cov;
    double* out_cov_rot;
    float* out_stats;
};

// Lane-batched: blockIdx.z = pipeline lane (independent sequence); per-keypoint tables are [lanes, cap, .] with
// cnt.n[lane] live rows, depth maps [lanes, H, W], rot [lanes, 9].  lanes = 1, cap = N is the plain call.
__global__ __launch_bounds__(256) void match_cov_kernel(CovSet s0, CovSet s1, mvMatchCovParams p, int cap, mvLaneCounts cnt) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int pl = blockIdx.z;
    if (n >= cnt.n[pl]) return;  // whole wave exits together
    const CovSet& S = blockIdx.y ? s1 : s0;   // the frame's two keypoint sets (kp0 on depth0, kp1 on depth1) share a launch
    const size_t ln = (size_t)pl * cap;
    const float* __restrict__ depth_map = S.depth_map + (size_t)pl * p.H * p.W;
    const float* __restrict__ kp_uv = S.kp_uv + 2 * ln;
    float* __restrict__ flow_cov = S.flow_cov + 3 * ln;
    const float* __restrict__ depth_cov = S.depth_cov ? S.depth_cov + ln : nullptr;
    const double* __restrict__ rot = S.rot ? S.rot + 9 * pl : nullptr;
    double* __restrict__ out_cov = S.out_cov + 9 * ln;
    double* __restrict__ out_cov_rot = S.out_cov_rot ? S.out_cov_rot + 9 * ln : nullptr;
    float* __restrict__ out_stats = S.out_stats ? S.out_stats + 2 * ln : nullptr;

    const float u = kp_uv[2 * n], v = kp_uv[2 * n + 1];
    const int iu = (int)u, iv = (int)v;  // .long(): truncation toward zero
    float suu = flow_cov[3 * n], svv = flow_cov[3 * n + 1];
    const float suv = flow_cov[3 * n + 2];
    suu = clamp_min_nanprop(suu, p.min_flow_cov_sq);
    svv = clamp_min_nanprop(svv, p.min_flow_cov_sq);
    if (lane == 0) {  // the reference clamps the caller's tensor in place
        flow_cov[3 * n] = suu;
        flow_cov[3 * n + 1] = svv;
    }

    // Sigma^-1 (torch.pinverse == inverse for a non-singular 2x2) and the 1/(2*pi*sqrt(det)) factor
    const float det = suu * svv - suv * suv;
    const float m00 = -0.5f * (svv / det), m01 = -0.5f * (-suv / det), m11 = -0.5f * (suu / det);
    const float cnorm = (2.f * 3.14159265358979323846f) * sqrtf(det);

    const int K = p.kernel_size, h = K >> 1, taps = K * K;
    float z[MAX_TAPS_PER_LANE], kv[MAX_TAPS_PER_LANE];
    float ksum = 0.f;
#pragma unroll
    for (int r = 0; r < MAX_TAPS_PER_LANE; ++r) {
        const int idx = r * 64 + lane;
        z[r] = 0.f;
        kv[r] = 0.f;
        if (idx < taps) {
            const int a = idx / K, b = idx - a * K;
            int yy = iv + (a - h), xx = iu + (b - h);
            // keypoints are >= mask_width from the border so the patch is always inside; clamp defensively
            yy = min(max(yy, 0), p.H - 1);
            xx = min(max(xx, 0), p.W - 1);
            z[r] = depth_map[yy * p.W + xx];
            const float x0 = (float)(a - h), x1 = (float)(b - h);
            const float q = (x0 * m00) * x0 + 2.f * ((x0 * m01) * x1) + (x1 * m11) * x1;
            kv[r] = expf(q) / cnorm;
            ksum += kv[r];
        }
    }
    ksum = wave_sum(ksum);

    float mu = 0.f;
#pragma unroll
    for (int r = 0; r < MAX_TAPS_PER_LANE; ++r) {
        kv[r] = kv[r] / ksum;
        mu += kv[r] * z[r];
    }
    mu = wave_sum(mu);

    float var;
    if (p.use_patch_var) {
        var = 0.f;
#pragma unroll
        for (int r = 0; r < MAX_TAPS_PER_LANE; ++r) {
            const float d = z[r] - mu;
            var += kv[r] * (d * d);
        }
        var = wave_sum(var);
    } else {
        var = depth_cov[n];
    }
    var = clamp_min_nanprop(var, p.min_depth_cov);

    if (lane == 0) {
        // Covariance_2to3_full, fp32, reference op order
        const float du = u - p.cx, dv = v - p.cy;
        const float fx2 = p.fx * p.fx, fy2 = p.fy * p.fy, fxy = p.fx * p.fy;
        const float d2 = mu * mu;
        const float sxx = (((du * du) * var) + (d2 * suu) + (suu * var)) / fx2;
        const float syy = (((dv * dv) * var) + (d2 * svv) + (svv * var)) / fy2;
        const float szz = var;
        const float sxy = (((du * dv) * var) + (d2 + var) * suv) / fxy;
        const float sxz = (var * du) / p.fx;
        const float syz = (var * dv) / p.fy;
        double c[9] = {szz, sxz, syz, sxz, sxx, sxy, syz, sxy, syy};
        double* o = out_cov + (size_t)n * 9;
#pragma unroll
        for (int i = 0; i < 9; ++i) o[i] = c[i];
        if (out_cov_rot && rot) {
            double R[9], t[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) R[i] = rot[i];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    t[3 * i + j] = (R[3 * i] * c[j] + R[3 * i + 1] * c[3 + j]) + R[3 * i + 2] * c[6 + j];
            double* orot = out_cov_rot + (size_t)n * 9;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    orot[3 * i + j] = (t[3 * i] * R[3 * j] + t[3 * i + 1] * R[3 * j + 1]) + t[3 * i + 2] * R[3 * j + 2];
        }
        if (out_stats) {
            out_stats[2 * n] = mu;
            out_stats[2 * n + 1] = var;
        }
    }
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
