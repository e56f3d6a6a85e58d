// A17-A22 — the per-point / per-pose fp64 arithmetic of the two-frame pose-graph solve, shared by
// pgo_solve.hip (device) and by tests/c_abi/pgo_twin.cpp (g++, host: a lane-by-lane replay of the kernel that the CPU suite
// checks against the oracle and the reference golden BEFORE the kernel ever reaches a GPU — test infrastructure only).
//
// Replaces (for the newest-frame pose, the only variable, Graphs.py:83):
//   residual graphs + analytic J      Module/Optimization/TwoFramePGO/Graphs.py:33-231
//   LM_analytic.step                  Module/Optimization/PyposeOptimizers.py:160-194
//   PyPose 0.6.8 Huber / FastTriggs / RobustModel.loss / TrustRegion / PINV / SE3 add_
//
// Round 4: every multiply-add of the hot path is an explicit fma() — the library is built with -ffp-contract=off (the fp32
// epilogues replay torch's one-rounding-per-op order), which left this fp64 code as separate v_mul_f64 / v_add_f64 pairs: 3.6 k
// of the DISP kernel's 9.4 k instructions.  One v_fma_f64 per pair halves the instruction stream of a kernel whose cost, alone
// and beside the volume GEMM, is its dependent instruction count.  Results move by fp64 roundoff (the reference's own matmuls
// are BLAS FMA chains in another order); the golden step / reject counts and the 1e-8 pose parity are what the tests hold.
#pragma once
#include <math.h>
#include <stdint.h>
#include "macvo_hip.h"

#if defined(__HIPCC__)
#define MV_HD __device__ __forceinline__
#define MV_UNROLL _Pragma("unroll")
#else
#define MV_HD static inline
#define MV_UNROLL
#endif

namespace pgo {

struct PgoArgs {
    const int32_t* offsets;
    const float* init_pose;
    const float* intrinsics;
    const float* baseline;
    const float* pos_Tw;
    const double* cov_Tw;
    const float* pixel2_uv;
    const float* pixel2_d;
    const float* pixel2_disp;
    const float* pixel2_disp_cov;
    const float* pixel2_uv_cov;
    const double* obs2_covTc;
    const uint8_t* valid;
    int min_points;
    double* out_pose;
    double* out_info;
    float* out_pose_f32;
    int spec;   // speculative reject rounds: 1 on, 0 off (MV_PGO_SPEC), 2 = on + round / trial counts into out_info[3] (debugging)
    // mv_pgo_solve_posed: the pose-dependent remainder of the backend folded into this launch (all null / zero otherwise).  The first
    // apply_live[prob] rows of a problem are rotated into the world frame with the problem's init_pose before any row is read:
    // pos_Tw = T p_cam, cov_Tw = R cov_Tc R^T, rot = R (pose_apply_dev.h); pose_sink: a second fp32 copy of the optimised pose
    const float* apply_pos_Tc;
    const double* apply_cov_Tc;
    float* apply_pos_Tw;
    double* apply_cov_Tw;
    double* apply_rot;
    float* pose_sink;
    int32_t apply_live[MV_MAX_LANES];
    const int32_t* live_dev;   // device-driven frame (mv_pgo_solve_posed_dev): the live-row count of problem l is live_dev[l * live_stride] in device memory
    int live_stride;           // (written by the backend's front launch on the same stream) instead of apply_live[l]
    // ... and the observation filters of the problem's rows (obs_filter_dev.h) in front of that: filter_flags >= 0 -> valid_out / count_out written
    int filter_flags;
    float filter_min_depth, filter_max_depth;
    int filter_cap;
    const uint8_t* filter_inbound;
    const float* filter_vals;
    uint8_t* valid_out;
    int32_t* count_out;
};

struct Pose {
    double t[3];
    double q[4];   // x y z w
    double R[9];   // row-major rotation matrix of q
    double ti[3];  // q^-1.Act(t): the translation of T^-1 up to sign (REPROJ / DISP residuals), once per pose instead of per point
};

struct Geometry {
    double fx, fy, cx, cy, blfx;
};

// 1: a step's first trial of the lean form BUILDS at the trial pose (28 values + the quality term in one reduction) instead of summing only its
// loss: an accepted trial — every step but a solve's last, in practice — then already is the next step's build pass.  0: the trial sums its loss
// and quality term alone and every step builds afresh (A/B; pgo_solve.hip and the host twin follow the same switch).
#ifndef MV_PGO_FUSED_BUILD
#define MV_PGO_FUSED_BUILD 1
#endif

constexpr int NRED = 55;    // full build: A_w 21 + g_w 6 + A_u 21 + g_u 6 + loss 1
constexpr int NLEAN = 28;   // lean build: A_w 21 + g_w 6 + loss 1
constexpr int NUNW = 27;    // the unweighted pair A_u 21 + g_u 6 on its own (first rejection of a step)

MV_HD double mv_rsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return rsqrt(x);
#else
    return 1.0 / sqrt(x);
#endif
}

// index of (j, k), j <= k, in the packed upper triangle of a 6x6
MV_HD constexpr int tri(int j, int k) { return j * 6 - (j * (j - 1)) / 2 + (k - j); }

// PyPose SO3_Act: p + w*uv + qv x uv with uv = 2 (qv x p)
MV_HD void quat_act(const double* q, const double* p, double* o) {
    double uv0 = fma(q[1], p[2], -(q[2] * p[1])), uv1 = fma(q[2], p[0], -(q[0] * p[2])), uv2 = fma(q[0], p[1], -(q[1] * p[0]));
    uv0 += uv0; uv1 += uv1; uv2 += uv2;
    o[0] = fma(q[1], uv2, fma(-q[2], uv1, fma(q[3], uv0, p[0])));
    o[1] = fma(q[2], uv0, fma(-q[0], uv2, fma(q[3], uv1, p[1])));
    o[2] = fma(q[0], uv1, fma(-q[1], uv0, fma(q[3], uv2, p[2])));
}

// R(q) and the translation of the inverse: everything a residual needs from the pose beyond (t, q)
MV_HD void pose_finish(Pose& P) {
    const double x = P.q[0], y = P.q[1], z = P.q[2], w = P.q[3];
    P.R[0] = fma(-2.0, fma(y, y, z * z), 1.0); P.R[1] = 2.0 * fma(x, y, -(z * w));        P.R[2] = 2.0 * fma(x, z, y * w);
    P.R[3] = 2.0 * fma(x, y, z * w);           P.R[4] = fma(-2.0, fma(x, x, z * z), 1.0); P.R[5] = 2.0 * fma(y, z, -(x * w));
    P.R[6] = 2.0 * fma(x, z, -(y * w));        P.R[7] = 2.0 * fma(y, z, x * w);           P.R[8] = fma(-2.0, fma(x, x, y * y), 1.0);
    const double qi[4] = {-x, -y, -z, w};
    quat_act(qi, P.t, P.ti);
}

// T <- Exp([rho, phi]) * T  (PyPose se3_Exp: t = Jl(phi) rho, q = so3_Exp(phi); SE3_Mul)
MV_HD void se3_left_update(Pose& P, const double* D) {
    const double eps = 2.220446049250313e-16;
    const double rho[3] = {D[0], D[1], D[2]}, phi[3] = {D[3], D[4], D[5]};
    const double th2 = fma(phi[0], phi[0], fma(phi[1], phi[1], phi[2] * phi[2]));
    double c1, c2, imag, real;
    if (th2 < 1.0e-2) {
        // |phi| < 0.1 rad (every LM step but a wild first one): Taylor series, truncation error < 3e-16 relative —
        // below the cancellation noise of PyPose's own closed forms at these angles — and no fp64 sin/cos calls
        const double h2 = 0.25 * th2;  // (theta/2)^2
        c1 = fma(-th2, fma(-th2, fma(-th2, fma(-th2, 1.0 / 3628800.0, 1.0 / 40320.0), 1.0 / 720.0), 1.0 / 24.0), 0.5);
        c2 = fma(-th2, fma(-th2, fma(-th2, fma(-th2, 1.0 / 39916800.0, 1.0 / 362880.0), 1.0 / 5040.0), 1.0 / 120.0), 1.0 / 6.0);
        imag = 0.5 * fma(-h2, fma(-h2, fma(-h2, fma(-h2, 1.0 / 362880.0, 1.0 / 5040.0), 1.0 / 120.0), 1.0 / 6.0), 1.0);
        real = fma(-h2, fma(-h2, fma(-h2, fma(-h2, fma(-h2, 1.0 / 3628800.0, 1.0 / 40320.0), 1.0 / 720.0), 1.0 / 24.0), 0.5), 1.0);
    } else {
        const double th = sqrt(th2);
        if (th > eps) {
            c1 = (1.0 - cos(th)) / th2;
            c2 = (th - sin(th)) / (th * th2);
            imag = sin(0.5 * th) / th;
            real = cos(0.5 * th);
        } else {
            const double th4 = th2 * th2;
            c1 = 0.5 - th2 / 24.0;
            c2 = 1.0 / 6.0 - th2 / 120.0;
            imag = 0.5 - th2 / 48.0 + th4 / 3840.0;
            real = 1.0 - th2 / 8.0 + th4 / 384.0;
        }
    }
    // Jl rho = rho + c1 (phi x rho) + c2 (phi x (phi x rho))
    const double k1[3] = {fma(phi[1], rho[2], -(phi[2] * rho[1])), fma(phi[2], rho[0], -(phi[0] * rho[2])), fma(phi[0], rho[1], -(phi[1] * rho[0]))};
    const double k2[3] = {fma(phi[1], k1[2], -(phi[2] * k1[1])), fma(phi[2], k1[0], -(phi[0] * k1[2])), fma(phi[0], k1[1], -(phi[1] * k1[0]))};
    const double te[3] = {fma(c2, k2[0], fma(c1, k1[0], rho[0])), fma(c2, k2[1], fma(c1, k1[1], rho[1])), fma(c2, k2[2], fma(c1, k1[2], rho[2]))};
    const double qe[4] = {phi[0] * imag, phi[1] * imag, phi[2] * imag, real};
    // t' = te + qe.Act(t);  q' = qe * q
    double rt[3];
    quat_act(qe, P.t, rt);
    const double a[3] = {qe[0], qe[1], qe[2]}, aw = qe[3];
    const double b[3] = {P.q[0], P.q[1], P.q[2]}, bw = P.q[3];
    const double nq[4] = {fma(aw, b[0], fma(bw, a[0], fma(a[1], b[2], -(a[2] * b[1])))),
                          fma(aw, b[1], fma(bw, a[1], fma(a[2], b[0], -(a[0] * b[2])))),
                          fma(aw, b[2], fma(bw, a[2], fma(a[0], b[1], -(a[1] * b[0])))),
                          fma(aw, bw, -fma(a[0], b[0], fma(a[1], b[1], a[2] * b[2])))};
    P.t[0] = te[0] + rt[0]; P.t[1] = te[1] + rt[1]; P.t[2] = te[2] + rt[2];
    P.q[0] = nq[0]; P.q[1] = nq[1]; P.q[2] = nq[2]; P.q[3] = nq[3];
    pose_finish(P);
}

// pp.optim.kernel.Huber on a squared norm, given its square root
MV_HD double huber_sn(double x, double sx, double delta) { return (sx < delta) ? x : fma(2.0 * delta, sx, -(delta * delta)); }

// torch.linalg.pinv of the symmetric 2x2 [[a, c], [c, b]] (+ optional independent third singular value s3
// of the block-diagonal 3x3) with relative cutoff rcond * sigma_max.
MV_HD void pinv_sym2_blk(double a, double b, double c, double s3, bool has3, double rcond, double& w00, double& w01, double& w11,
                         double& w22) {
    const double tr = a + b, df = a - b;
    const double rad = sqrt(0.25 * df * df + c * c);
    const double l1 = 0.5 * tr + rad, l2 = 0.5 * tr - rad;
    double smax = fmax(fabs(l1), fabs(l2));
    if (has3) smax = fmax(smax, fabs(s3));
    const double cut = rcond * smax;
    const bool k1 = fabs(l1) > cut, k2 = fabs(l2) > cut;
    if (k1 && k2) {
        const double det = a * b - c * c;
        w00 = b / det; w01 = -c / det; w11 = a / det;
    } else if (k1 || k2) {
        const double l = k1 ? l1 : l2, lo = k1 ? l2 : l1;
        const double s = 1.0 / (l * (l - lo));  // (A - lo I) / (l - lo) is the projector onto l's eigenvector
        w00 = (a - lo) * s; w01 = c * s; w11 = (b - lo) * s;
    } else {
        w00 = w01 = w11 = 0.0;
    }
    w22 = (has3 && fabs(s3) > cut) ? 1.0 / s3 : 0.0;
}

// general 3x3 inverse by cofactors (== torch.pinverse for the well-conditioned fp64 covariances of the ICP graph)
MV_HD void inv3(const double* m, double* o) {
    const double c00 = fma(m[4], m[8], -(m[5] * m[7])), c01 = fma(m[5], m[6], -(m[3] * m[8])), c02 = fma(m[3], m[7], -(m[4] * m[6]));
    const double det = fma(m[0], c00, fma(m[1], c01, m[2] * c02));
    const double id = 1.0 / det;
    o[0] = c00 * id; o[1] = fma(m[2], m[7], -(m[1] * m[8])) * id; o[2] = fma(m[1], m[5], -(m[2] * m[4])) * id;
    o[3] = c01 * id; o[4] = fma(m[0], m[8], -(m[2] * m[6])) * id; o[5] = fma(m[2], m[3], -(m[0] * m[5])) * id;
    o[6] = c02 * id; o[7] = fma(m[1], m[6], -(m[0] * m[7])) * id; o[8] = fma(m[0], m[4], -(m[1] * m[3])) * id;
}

// Everything a point contributes, gathered once (fp32 buffers widened to fp64 exactly as the reference's
// `.to(torch.double)` does, Optimizer.py:84-85).
template <int GT>
struct PointData {
    bool valid;
    double pw[3];      // pos_Tw
    double obs[3];     // REPROJ/DISP: (u, v, disparity) ; ICP: points_Tc (pixel2point_NED evaluated in fp32)
    double W[4];       // REPROJ/DISP: pinv(Sigma_i) = [[W0, W1, 0], [W1, W2, 0], [0, 0, W3]] (constant); ICP: unused
    double So[9], Sp[9];  // ICP only: obs2_covTc, cov_Tw
};

template <int GT>
MV_HD void load_point(const PgoArgs& a, const Geometry& g, const mvLMParams& lm, int i, bool in_range, PointData<GT>& d) {
    d.valid = in_range && (a.valid ? (a.valid[i] != 0) : true);
    if (!d.valid) return;
    d.pw[0] = (double)a.pos_Tw[3 * i]; d.pw[1] = (double)a.pos_Tw[3 * i + 1]; d.pw[2] = (double)a.pos_Tw[3 * i + 2];
    if (GT == MV_GRAPH_ICP) {
        // points_Tc = pixel2point_NED(pixel2_uv, pixel2_d, K) built in fp32 (Graphs.py:49-51), then cast
        const float u = a.pixel2_uv[2 * i], v = a.pixel2_uv[2 * i + 1], dd = a.pixel2_d[i];
        d.obs[0] = (double)dd;
        d.obs[1] = (double)(((u - (float)g.cx) * dd) / (float)g.fx);
        d.obs[2] = (double)(((v - (float)g.cy) * dd) / (float)g.fy);
        MV_UNROLL
        for (int k = 0; k < 9; ++k) { d.So[k] = a.obs2_covTc[9 * (size_t)i + k]; d.Sp[k] = a.cov_Tw[9 * (size_t)i + k]; }
    } else {
        d.obs[0] = (double)a.pixel2_uv[2 * i]; d.obs[1] = (double)a.pixel2_uv[2 * i + 1];
        const double suu = (double)a.pixel2_uv_cov[3 * i], svv = (double)a.pixel2_uv_cov[3 * i + 1],
                     suv = (double)a.pixel2_uv_cov[3 * i + 2];
        if (GT == MV_GRAPH_DISP) {
            d.obs[2] = (double)a.pixel2_disp[i];
            pinv_sym2_blk(suu, svv, suv, (double)a.pixel2_disp_cov[i], true, lm.pinv_rcond, d.W[0], d.W[1], d.W[2], d.W[3]);
        } else {
            d.obs[2] = 0.0;
            pinv_sym2_blk(suu, svv, suv, 0.0, false, lm.pinv_rcond, d.W[0], d.W[1], d.W[2], d.W[3]);
        }
    }
}

// residual block under pose P; returns |r|^2.  pc = T*p_c (ICP) or p_c = T^-1 p_w (REPROJ/DISP); ix = 1 / p_c.x (REPROJ/DISP).
template <int GT>
MV_HD double residual(const Geometry& g, const Pose& P, const double* pw, const double* obs, double* r, double* pc, double& ix) {
    if (GT == MV_GRAPH_ICP) {
        double rp[3];
        quat_act(P.q, obs, rp);
        pc[0] = rp[0] + P.t[0]; pc[1] = rp[1] + P.t[1]; pc[2] = rp[2] + P.t[2];
        r[0] = pc[0] - pw[0]; r[1] = pc[1] - pw[1]; r[2] = pc[2] - pw[2];
        ix = 0.0;
        return fma(r[0], r[0], fma(r[1], r[1], r[2] * r[2]));
    } else {
        // p_c = T^-1 p_w : Inv = (-q^-1.Act(t), q^-1), Act = q^-1.Act(p_w) + t_inv
        const double qi[4] = {-P.q[0], -P.q[1], -P.q[2], P.q[3]};
        double rp[3];
        quat_act(qi, pw, rp);
        pc[0] = rp[0] - P.ti[0]; pc[1] = rp[1] - P.ti[1]; pc[2] = rp[2] - P.ti[2];
        // point2pixel_NED = homo2cart(p_EDN K^T): u = (fx Y + cx X) / X, v = (fy Z + cy X) / X  (homo2cart clamps |X| away from 0)
        const double X = pc[0];
        double den = fmax(fabs(X), 2.2250738585072014e-308);
        den = (X >= 0.0) ? den : -den;
        const double iden = 1.0 / den;
        ix = 1.0 / X;
        r[0] = fma(fma(g.fx, pc[1], g.cx * X), iden, -obs[0]);
        r[1] = fma(fma(g.fy, pc[2], g.cy * X), iden, -obs[1]);
        double n2 = fma(r[0], r[0], r[1] * r[1]);
        if (GT == MV_GRAPH_DISP) {
            r[2] = fma(ix, g.blfx, -obs[2]);
            n2 = fma(r[2], r[2], n2);
        } else {
            r[2] = 0.0;
        }
        return n2;
    }
}

// RobustModel.loss contribution of one point under pose P
template <int GT>
MV_HD double point_loss(const Geometry& g, const mvLMParams& lm, const Pose& P, const double* pw, const double* obs) {
    double r[3], pc[3], ix;
    const double n2 = residual<GT>(g, P, pw, obs, r, pc, ix);
    return huber_sn(n2, sqrt(n2), lm.huber_delta);
}

// What a build pass keeps per point for the trust-region quality of the step's first trial (lean form): the FastTriggs-scaled
// Jacobian rows and residual, i.e. exactly the J, R that TrustRegion.update receives.
struct PointLin {
    double J[3][6];   // row 2 unused (zero) for REPROJ
    double r[3];
    double s2;        // rho'(|r|^2): J and R are both scaled by its square root => s2 on every product
};

// One point under pose P: residual, analytic Jacobian, information block.  Always: loss -> acc[NLEAN - 1 or NRED - 1],
// A_w = J^T W J (21), g_w = J^T W r (6).  FULL (several points per thread): ADDS to acc, also the unweighted A_u (21), g_u (6) (layout
// of the 55-value build); !FULL (one point per thread): WRITES acc, and J, r, s2 are returned in `lin` instead (28-value build).
template <int GT, bool FULL>
MV_HD void accumulate_point(const Geometry& g, const mvLMParams& lm, const Pose& P, const PointData<GT>& d, double* acc, PointLin& lin) {
    constexpr int NR = (GT == MV_GRAPH_REPROJ) ? 2 : 3;
    auto put = [](double& slot, double v) { if (FULL) slot += v; else slot = v; };
    double r[3], pc[3], ix;
    const double n2 = residual<GT>(g, P, d.pw, d.obs, r, pc, ix);
    const double sn = sqrt(n2);
    put(acc[FULL ? NRED - 1 : NLEAN - 1], huber_sn(n2, sn, lm.huber_delta));
    // FastTriggs: s = sqrt(rho'(|r|^2)); both R and J are scaled by s => s^2 on every product
    const double s2 = (sn < lm.huber_delta) ? 1.0 : (lm.huber_delta / sn);

    double J[3][6];
    double WJ[3][6], Wr[3];
    if (GT == MV_GRAPH_ICP) {
        // J = [I, -skew(T p_c)]
        MV_UNROLL
        for (int rr = 0; rr < 3; ++rr)
            MV_UNROLL
            for (int c = 0; c < 6; ++c) J[rr][c] = 0.0;
        J[0][0] = J[1][1] = J[2][2] = 1.0;
        J[0][4] = pc[2];  J[0][5] = -pc[1];
        J[1][3] = -pc[2]; J[1][5] = pc[0];
        J[2][3] = pc[1];  J[2][4] = -pc[0];
        // Sigma_i = R Sigma_obs R^T + Sigma_pt ; W_i = pinv(Sigma_i)   (Graphs.py:62-68, Optimizer.py:96-98)
        double T1[9], S[9], Wi[9];
        MV_UNROLL
        for (int x = 0; x < 3; ++x)
            MV_UNROLL
            for (int y = 0; y < 3; ++y)
                T1[3 * x + y] = fma(P.R[3 * x + 2], d.So[6 + y], fma(P.R[3 * x + 1], d.So[3 + y], P.R[3 * x] * d.So[y]));
        MV_UNROLL
        for (int x = 0; x < 3; ++x)
            MV_UNROLL
            for (int y = 0; y < 3; ++y)
                S[3 * x + y] = fma(T1[3 * x + 2], P.R[3 * y + 2], fma(T1[3 * x + 1], P.R[3 * y + 1], T1[3 * x] * P.R[3 * y])) + d.Sp[3 * x + y];
        inv3(S, Wi);
        MV_UNROLL
        for (int x = 0; x < 3; ++x) {
            Wr[x] = fma(Wi[3 * x + 2], r[2], fma(Wi[3 * x + 1], r[1], Wi[3 * x] * r[0]));
            MV_UNROLL
            for (int c = 0; c < 6; ++c) WJ[x][c] = fma(Wi[3 * x + 2], J[2][c], fma(Wi[3 * x + 1], J[1][c], Wi[3 * x] * J[0][c]));
        }
    } else {
        // G = d p_c / d delta = [-R^T, R^T skew(p_w)]   (3 x 6)
        double G[3][6];
        MV_UNROLL
        for (int x = 0; x < 3; ++x) {
            const double rt0 = P.R[x], rt1 = P.R[3 + x], rt2 = P.R[6 + x];  // row x of R^T
            G[x][0] = -rt0; G[x][1] = -rt1; G[x][2] = -rt2;
            // R^T skew(p): col0 = R^T (0, pz, -py), col1 = R^T (-pz, 0, px), col2 = R^T (py, -px, 0)
            G[x][3] = fma(rt1, d.pw[2], -(rt2 * d.pw[1]));
            G[x][4] = fma(rt2, d.pw[0], -(rt0 * d.pw[2]));
            G[x][5] = fma(rt0, d.pw[1], -(rt1 * d.pw[0]));
        }
        // d pixel / d p_c: [[-fx Y / X^2, fx / X, 0], [-fy Z / X^2, 0, fy / X]], d disparity / d p_c = [-bl fx / X^2, 0, 0]
        const double ix2 = ix * ix;
        const double j01 = g.fx * ix, j12 = g.fy * ix;
        const double j00 = -(j01 * pc[1]) * ix, j10 = -(j12 * pc[2]) * ix;
        MV_UNROLL
        for (int c = 0; c < 6; ++c) {
            J[0][c] = fma(j00, G[0][c], j01 * G[1][c]);
            J[1][c] = fma(j10, G[0][c], j12 * G[2][c]);
            J[2][c] = 0.0;
        }
        if (GT == MV_GRAPH_DISP) {
            const double jd = -(g.blfx * ix2);
            MV_UNROLL
            for (int c = 0; c < 6; ++c) J[2][c] = jd * G[0][c];
        }
        // W = [[W0, W1, 0], [W1, W2, 0], [0, 0, W3]]
        Wr[0] = fma(d.W[0], r[0], d.W[1] * r[1]);
        Wr[1] = fma(d.W[1], r[0], d.W[2] * r[1]);
        Wr[2] = (GT == MV_GRAPH_DISP) ? d.W[3] * r[2] : 0.0;
        MV_UNROLL
        for (int c = 0; c < 6; ++c) {
            WJ[0][c] = fma(d.W[0], J[0][c], d.W[1] * J[1][c]);
            WJ[1][c] = fma(d.W[1], J[0][c], d.W[2] * J[1][c]);
            WJ[2][c] = (GT == MV_GRAPH_DISP) ? d.W[3] * J[2][c] : 0.0;
        }
    }
    // reference: J_T = J^T @ weight ; A = J_T @ J ; b = -J_T @ R   (PyposeOptimizers.py:170-176), J and R FastTriggs-scaled
    double sJ[3][6];
    MV_UNROLL
    for (int x = 0; x < NR; ++x)
        MV_UNROLL
        for (int c = 0; c < 6; ++c) sJ[x][c] = s2 * J[x][c];
    MV_UNROLL
    for (int j = 0; j < 6; ++j) {
        double gwj = sJ[0][j] * Wr[0];
        MV_UNROLL
        for (int x = 1; x < NR; ++x) gwj = fma(sJ[x][j], Wr[x], gwj);
        put(acc[21 + j], gwj);
        MV_UNROLL
        for (int k = j; k < 6; ++k) {
            double aw = sJ[0][j] * WJ[0][k];
            MV_UNROLL
            for (int x = 1; x < NR; ++x) aw = fma(sJ[x][j], WJ[x][k], aw);
            put(acc[tri(j, k)], aw);
        }
    }
    if (FULL) {
        MV_UNROLL
        for (int j = 0; j < 6; ++j) {
            double guj = sJ[0][j] * r[0];
            MV_UNROLL
            for (int x = 1; x < NR; ++x) guj = fma(sJ[x][j], r[x], guj);
            put(acc[48 + j], guj);
            MV_UNROLL
            for (int k = j; k < 6; ++k) {
                double au = sJ[0][j] * J[0][k];
                MV_UNROLL
                for (int x = 1; x < NR; ++x) au = fma(sJ[x][j], J[x][k], au);
                put(acc[27 + tri(j, k)], au);
            }
        }
    } else {
        MV_UNROLL
        for (int x = 0; x < 3; ++x) {
            MV_UNROLL
            for (int c = 0; c < 6; ++c) lin.J[x][c] = J[x][c];
            lin.r[x] = r[x];
        }
        lin.s2 = s2;
    }
}

// the unweighted pair of the lean form, from what the build pass kept: u[0..20] = s2 J^T J, u[21..26] = s2 J^T r
template <int GT>
MV_HD void unweighted_point(const PointLin& lin, double* u) {
    constexpr int NR = (GT == MV_GRAPH_REPROJ) ? 2 : 3;
    double sJ[3][6];
    MV_UNROLL
    for (int x = 0; x < NR; ++x)
        MV_UNROLL
        for (int c = 0; c < 6; ++c) sJ[x][c] = lin.s2 * lin.J[x][c];
    MV_UNROLL
    for (int j = 0; j < 6; ++j) {
        double guj = sJ[0][j] * lin.r[0];
        MV_UNROLL
        for (int x = 1; x < NR; ++x) guj = fma(sJ[x][j], lin.r[x], guj);
        u[21 + j] = guj;
        MV_UNROLL
        for (int k = j; k < 6; ++k) {
            double au = sJ[0][j] * lin.J[0][k];
            MV_UNROLL
            for (int x = 1; x < NR; ++x) au = fma(sJ[x][j], lin.J[x][k], au);
            u[tri(j, k)] = au;
        }
    }
}

// one point's term of (J D)^T (2 R + J D) on the FastTriggs-scaled J, R: s2 * sum_x (J D)_x (2 r_x + (J D)_x)
template <int GT>
MV_HD double quality_point(const PointLin& lin, const double (&D)[6]) {
    constexpr int NR = (GT == MV_GRAPH_REPROJ) ? 2 : 3;
    double q = 0.0;
    MV_UNROLL
    for (int x = 0; x < NR; ++x) {
        double jd = lin.J[x][0] * D[0];
        MV_UNROLL
        for (int c = 1; c < 6; ++c) jd = fma(lin.J[x][c], D[c], jd);
        q = fma(jd, fma(2.0, lin.r[x], jd), q);
    }
    return lin.s2 * q;
}

// solve A D = b, b = -gw, by Cholesky (A = L L^T) with the diagonal of A taken from `dg6` (the damped one); every thread solves
// redundantly (uniform control flow).  One reciprocal square root per pivot (l_jj = dd / sqrt(dd) to ~1 ulp), the substitutions
// multiply by the reciprocal pivots.  Returns false where PyPose reports "Linear solver failed".
MV_HD bool chol_solve6(const double* Aw, const double (&dg6)[6], const double* gw, double (&D)[6]) {
    double L[6][6], linv[6];
    bool ok = true;
    MV_UNROLL
    for (int j = 0; j < 6; ++j) {
        double dd = dg6[j];
        MV_UNROLL
        for (int k = 0; k < j; ++k) dd = fma(-L[j][k], L[j][k], dd);
        ok = ok && (dd > 0.0) && (dd < INFINITY);
        const double inv = mv_rsqrt(dd);
        L[j][j] = dd * inv;
        linv[j] = inv;
        MV_UNROLL
        for (int i2 = j + 1; i2 < 6; ++i2) {
            double sacc = Aw[tri(j, i2)];
            MV_UNROLL
            for (int k = 0; k < j; ++k) sacc = fma(-L[i2][k], L[j][k], sacc);
            L[i2][j] = sacc * inv;
        }
    }
    if (!ok) return false;
    double yv[6];
    MV_UNROLL
    for (int j = 0; j < 6; ++j) {
        double sacc = -gw[j];
        MV_UNROLL
        for (int k = 0; k < j; ++k) sacc = fma(-L[j][k], yv[k], sacc);
        yv[j] = sacc * linv[j];
    }
    MV_UNROLL
    for (int j = 5; j >= 0; --j) {
        double sacc = yv[j];
        MV_UNROLL
        for (int k = j + 1; k < 6; ++k) sacc = fma(-L[k][j], D[k], sacc);
        D[j] = sacc * linv[j];
    }
    return true;
}

// TrustRegion.update: quality = (last - loss) / -((J D)^T (2 R + J D)) on the corrected, unweighted J, R, from the reduced pair
MV_HD double tr_quality(const double (&D)[6], const double* gu, const double* Au, double last, double loss) {
    double dAd = 0.0, dg = 0.0;
    MV_UNROLL
    for (int j = 0; j < 6; ++j) {
        dg = fma(D[j], gu[j], dg);
        double row = 0.0;
        MV_UNROLL
        for (int k = 0; k < 6; ++k) row = fma(D[k], Au[(j <= k) ? tri(j, k) : tri(k, j)], row);
        dAd = fma(D[j], row, dAd);
    }
    return (last - loss) / -fma(2.0, dg, dAd);
}

// ... and the radius / damping update it drives; `branch`: 1 quality > high, 2 > low, 3 shrink
MV_HD void tr_apply(const mvLMParams& lm, int branch, double& damping, double& tr_down) {
    double radius = 1.0 / damping;
    if (branch == 1) {
        radius = lm.tr_up * radius;
        tr_down = lm.tr_down;
    } else if (branch == 2) {
        tr_down = lm.tr_down;
    } else {
        radius = radius * tr_down;
        tr_down = tr_down * lm.tr_factor;
    }
    tr_down = fmax(lm.tr_min, fmin(tr_down, lm.tr_max));
    radius = fmax(lm.tr_min, fmin(radius, lm.tr_max));
    damping = 1.0 / radius;
}
MV_HD int tr_update(const mvLMParams& lm, double quality, double& damping, double& tr_down) {
    const int branch = (quality > lm.tr_high) ? 1 : (quality > lm.tr_low) ? 2 : 3;
    tr_apply(lm, branch, damping, tr_down);
    return branch;
}

}  // namespace pgo
