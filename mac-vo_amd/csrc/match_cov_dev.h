// Device body of the MAC-VO 2D->3D covariance model (one 64-lane wave per keypoint), shared by match_cov_kernel (match_cov.hip) and the
// fused backend kernel (frontend_ops.hip: gather + track + back-projection + both covariances + observation filter in one launch) so that
// both produce the same bits.  See match_cov.hip for the reference citations and the design notes.
#pragma once
#include "common.h"
#include <math.h>

namespace mvcov {

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
// One wave = one keypoint `n` of pipeline lane `pl` of one keypoint set S (the whole wave must call it together).
// (u, v) = the keypoint, (suu, svv, suv) = its match covariance as the tables hold them: match_cov_wave reads them from S.kp_uv / S.flow_cov,
// the fused backend kernel hands over the values it has just computed (another wave may still be writing those tables).
__device__ __forceinline__ void match_cov_wave_vals(const CovSet& S, const mvMatchCovParams& p, int cap, int pl, int n, float u, float v,
                                                    float suu, float svv, float suv) {
    const int lane = threadIdx.x & 63;
    const size_t ln = (size_t)pl * cap;
    const float* __restrict__ depth_map = S.depth_map + (size_t)pl * p.H * p.W;
    float* __restrict__ flow_cov = S.flow_cov + 3 * ln;
    const float* __restrict__ depth_cov = S.depth_cov ? S.depth_cov + ln : nullptr;
    const double* __restrict__ rot = S.rot ? S.rot + 9 * pl : nullptr;
    double* __restrict__ out_cov = S.out_cov + 9 * ln;
    double* __restrict__ out_cov_rot = S.out_cov_rot ? S.out_cov_rot + 9 * ln : nullptr;
    float* __restrict__ out_stats = S.out_stats ? S.out_stats + 2 * ln : nullptr;

    const int iu = (int)u, iv = (int)v;  // .long(): truncation toward zero
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


__device__ __forceinline__ void match_cov_wave(const CovSet& S, const mvMatchCovParams& p, int cap, int pl, int n) {
    const size_t ln = (size_t)pl * cap;
    const float* __restrict__ kp_uv = S.kp_uv + 2 * ln;
    const float* __restrict__ flow_cov = S.flow_cov + 3 * ln;
    match_cov_wave_vals(S, p, cap, pl, n, kp_uv[2 * n], kp_uv[2 * n + 1], flow_cov[3 * n], flow_cov[3 * n + 1], flow_cov[3 * n + 2]);
}

}  // namespace mvcov
