// Device body of the pose-dependent remainder of a frame's backend (Odometry/MACVO.py:273-281), shared by pose_apply_kernel (frontend_ops.hip)
// and the prologue of pgo_solve_kernel (pgo_solve.hip: mv_pgo_solve_posed folds the rotation into the world frame into the solve's launch) so
// that both produce the same bits:  pos_Tw = T_prev * p_cam (fp32, PyPose SE3 Act), rot = R_prev (fp32 -> fp64, as backproject_kernel
// writes it), cov_Tw = R cov_Tc R^T (fp64, the grouping of match_cov_kernel).
#pragma once
#include "common.h"

// pp.SO3.matrix() in the pose dtype (fp32): columns are SO3_Act(q, e_i)  (PyPose: self.Act(I).T)
__device__ __forceinline__ void quat_act_f32(const float* q, const float* p, float* o) {
    float uv0 = q[1] * p[2] - q[2] * p[1], uv1 = q[2] * p[0] - q[0] * p[2], uv2 = q[0] * p[1] - q[1] * p[0];
    uv0 += uv0; uv1 += uv1; uv2 += uv2;
    o[0] = (p[0] + q[3] * uv0) + (q[1] * uv2 - q[2] * uv1);
    o[1] = (p[1] + q[3] * uv1) + (q[2] * uv0 - q[0] * uv2);
    o[2] = (p[2] + q[3] * uv2) + (q[0] * uv1 - q[1] * uv0);
}

// R (row-major, fp64) of the fp32 pose's quaternion: columns are SO3_Act(q, e_i)
__device__ __forceinline__ void mv_pose_rotation(const float* __restrict__ pose, double* R) {
    const float q[4] = {pose[3], pose[4], pose[5], pose[6]};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float e[3] = {c == 0 ? 1.f : 0.f, c == 1 ? 1.f : 0.f, c == 2 ? 1.f : 0.f};
        float col[3];
        quat_act_f32(q, e, col);
        R[0 * 3 + c] = (double)col[0];
        R[1 * 3 + c] = (double)col[1];
        R[2 * 3 + c] = (double)col[2];
    }
}

// row n of a lane's tables (pointers already offset to the lane; pos_Tc / pos_Tw and cov / cov_rot may be null pairs)
__device__ __forceinline__ void mv_pose_apply_row(const float* __restrict__ pose, const double* R, int n, const float* pos_Tc, const double* cov,
                                                  float* pos_Tw, double* cov_rot) {
    const float t[3] = {pose[0], pose[1], pose[2]}, q[4] = {pose[3], pose[4], pose[5], pose[6]};
    if (pos_Tw && pos_Tc) {
        const float p[3] = {pos_Tc[3 * n], pos_Tc[3 * n + 1], pos_Tc[3 * n + 2]};
        float r[3];
        quat_act_f32(q, p, r);
        pos_Tw[3 * n] = r[0] + t[0]; pos_Tw[3 * n + 1] = r[1] + t[1]; pos_Tw[3 * n + 2] = r[2] + t[2];
    }
    if (cov_rot && cov) {
        double c[9], tm[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) c[i] = cov[(size_t)n * 9 + i];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) tm[3 * i + j] = (R[3 * i] * c[j] + R[3 * i + 1] * c[3 + j]) + R[3 * i + 2] * c[6 + j];
        double* o = cov_rot + (size_t)n * 9;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) o[3 * i + j] = (tm[3 * i] * R[3 * j] + tm[3 * i + 1] * R[3 * j + 1]) + tm[3 * i + 2] * R[3 * j + 2];
    }
}
