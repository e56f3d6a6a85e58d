// ROUND-3 FORM (fallback / A/B, MV_PGO_V1=1): A17-A22 — covariance-weighted two-frame pose-graph solve on SE(3), one 64-lane wave per problem
// (SURVEY.md §8 A17-A22).
//
// Replaces, for the newest-frame pose (the only variable, Graphs.py:83):
//   TwoFrame_PGO._optimize            Module/Optimization/TwoFramePGO/Optimizer.py:81-102
//   residual graphs + analytic J      Module/Optimization/TwoFramePGO/Graphs.py:33-231
//   LM_analytic.step                  Module/Optimization/PyposeOptimizers.py:160-194
//   PyPose 0.6.8 Huber / FastTriggs / RobustModel.loss / TrustRegion / PINV / StopOnPlateau / SE3 add_
//
// What the reference materialises and this kernel does not:
//   * the dense 3N x 3N block_diag weight (2.9 MB fp64 for N = 200, rebuilt every outer iteration):
//     here each point applies its own 3x3 / 2x2 information block in registers;
//   * J [3N,7] and the 7x600 @ 600x600 matmul: here J_i^T W_i J_i (21 unique entries), J_i^T W_i r_i (6),
//     plus the UNWEIGHTED J^T J (21) and J^T r (6) that TrustRegion's quality ratio needs
//     ((J D)^T (2R + J D) = 2 D^T J^T R + D^T J^T J D), are accumulated per thread, tree-reduced inside each
//     wavefront with DPP butterflies (quad_perm / row_half_mirror / row_mirror + 4 readlanes) and combined across
//     the workgroup's 4 waves through a 4 x 55 fp64 LDS table — 55 fp64 values per build pass;
//   * the dead 7th tangent column (clamped to 1e-6, b_7 = 0 => D_7 = 0) is dropped analytically;
//     the 6x6 SPD system is solved by an in-register Cholesky instead of an SVD pseudo-inverse
//     (identical up to fp64 roundoff whenever A is non-singular, which the diagonal clamp + multiplicative
//     damping guarantee).
// One 256-thread workgroup (4 waves) per problem: with N <= 256 every thread owns one point and keeps its
// observation, world point and information matrix in registers for the whole solve (nothing is re-read).
// The whole <=10-step LM loop (with the inner reject/damp loop) runs on the device: one launch per batch of
// problems, no host round trips.  Latency-bound for a single problem (report us/solve), throughput-bound
// for large batches (report solves/s).
#include "common.h"
#include <math.h>

namespace {

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
};

struct Pose {
    double t[3];
    double q[4];   // x y z w
    double R[9];   // row-major rotation matrix of q
};


// ---- fp64 wavefront sum with DPP (all lanes must be active); result is wave-uniform -----------------------
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, false);
    const long long o = ((long long)hi << 32) | (unsigned)lo;
    return v + __builtin_bit_cast(double, o);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)b, lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]  : xor 1
    v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]  : xor 2
    v = dpp_add<0x141>(v);  // row_half_mirror      : other quad of the 8-lane half (all 4 lanes already equal)
    v = dpp_add<0x140>(v);  // row_mirror           : other half of the 16-lane row
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

constexpr int NRED = 55;  // 21 + 6 + 21 + 6 + 1

// workgroup sum of `n` per-thread values: DPP inside the wave, LDS table across the NW waves; every thread gets all sums.
// NW = 1 (throughput variant, one wave per problem) needs no LDS and no barrier at all.
template <int N, int NW>
__device__ __forceinline__ void block_sum(double (&v)[N], double (*__restrict__ tab)[NRED]) {
    if (NW == 1) {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = wave_sum_dpp(v[k]);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double s = wave_sum_dpp(v[k]);
        if (lane == 0) tab[wave][k] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double s = tab[0][k];
#pragma unroll
        for (int w = 1; w < NW; ++w) s += tab[w][k];
        v[k] = s;
    }
    __syncthreads();
}

// The 55-value reduction of a build pass, round 3.  In-kernel cycle stamps (profiles/probes/pgo_stamps.py) put the form above — per
// value four dependent DPP stages (two v_mov_dpp + one v_add_f64 each, with their wait states), eight v_readlane and three more
// adds — at 13.7 k of the 30 k cycles of an LM step: fp64 has no DPP-fused add, and the readlane -> SGPR -> VALU round trip of 55
// values is a long dependent instruction stream for the single wave of a SIMD.  Here only the three cheapest stages stay in
// registers (quad_perm x2 + row_half_mirror: every 8-lane group then holds its sum); the 8 partials per wave and value go to LDS
// (one ds_write_b64 per value with 8 lanes active), 55 threads add the 32 partials of one value each (ds_read_b128, four
// independent accumulators), and every thread reads the 55 sums back as broadcast ds_read_b128.  14 KB of LDS, three barriers.
// The summation tree changes (8-lane groups first, then 32 partials in order), i.e. results move by ~1e-16 relative.
template <int NW>
__device__ __forceinline__ void block_sum_build(double (&v)[NRED], double* __restrict__ part /* [NRED][NW * 8] */,
                                                double* __restrict__ fin /* [NRED + 1] */) {
    static_assert(NW == 4, "the wide variant");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, t = threadIdx.x;
#pragma unroll
    for (int k = 0; k < NRED; ++k) {
        double s = v[k];
        s = dpp_add<0xB1>(s);    // quad_perm [1,0,3,2]
        s = dpp_add<0x4E>(s);    // quad_perm [2,3,0,1]
        s = dpp_add<0x141>(s);   // row_half_mirror: the other quad of the 8-lane half
        if ((lane & 7) == 0) part[k * (NW * 8) + wave * 8 + (lane >> 3)] = s;
    }
    __syncthreads();
    if (t < NRED) {
        const double* p = part + t * (NW * 8);
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int i = 0; i < NW * 8; i += 4) { a0 += p[i]; a1 += p[i + 1]; a2 += p[i + 2]; a3 += p[i + 3]; }
        fin[t] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NRED; ++k) v[k] = fin[k];
    __syncthreads();   // `part` / `fin` are rewritten by the next pass
}

__device__ __forceinline__ void quat_to_R(Pose& P) {
    const double x = P.q[0], y = P.q[1], z = P.q[2], w = P.q[3];
    P.R[0] = 1 - 2 * (y * y + z * z); P.R[1] = 2 * (x * y - z * w);     P.R[2] = 2 * (x * z + y * w);
    P.R[3] = 2 * (x * y + z * w);     P.R[4] = 1 - 2 * (x * x + z * z); P.R[5] = 2 * (y * z - x * w);
    P.R[6] = 2 * (x * z - y * w);     P.R[7] = 2 * (y * z + x * w);     P.R[8] = 1 - 2 * (x * x + y * y);
}

// PyPose SO3_Act: p + w*uv + qv x uv with uv = 2 (qv x p)
__device__ __forceinline__ void quat_act(const double* q, const double* p, double* o) {
    double uv0 = q[1] * p[2] - q[2] * p[1], uv1 = q[2] * p[0] - q[0] * p[2], uv2 = q[0] * p[1] - q[1] * p[0];
    uv0 += uv0; uv1 += uv1; uv2 += uv2;
    o[0] = p[0] + q[3] * uv0 + (q[1] * uv2 - q[2] * uv1);
    o[1] = p[1] + q[3] * uv1 + (q[2] * uv0 - q[0] * uv2);
    o[2] = p[2] + q[3] * uv2 + (q[0] * uv1 - q[1] * uv0);
}

// T <- Exp([rho, phi]) * T  (PyPose se3_Exp: t = Jl(phi) rho, q = so3_Exp(phi); SE3_Mul)
__device__ void se3_left_update(Pose& P, const double* D) {
    const double eps = 2.220446049250313e-16;
    const double rho[3] = {D[0], D[1], D[2]}, phi[3] = {D[3], D[4], D[5]};
    const double th2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
    const double th = sqrt(th2);
    double c1, c2, imag, real;
    if (th2 < 1.0e-2) {
        // |phi| < 0.1 rad (every LM step but a wild first one): Taylor series, truncation error < 3e-16 relative —
        // below the cancellation noise of PyPose's own closed forms at these angles — and no fp64 sin/cos calls
        const double h2 = 0.25 * th2;  // (theta/2)^2
        c1 = 0.5 - th2 * (1.0 / 24.0 - th2 * (1.0 / 720.0 - th2 * (1.0 / 40320.0 - th2 * (1.0 / 3628800.0))));
        c2 = 1.0 / 6.0 - th2 * (1.0 / 120.0 - th2 * (1.0 / 5040.0 - th2 * (1.0 / 362880.0 - th2 * (1.0 / 39916800.0))));
        imag = 0.5 * (1.0 - h2 * (1.0 / 6.0 - h2 * (1.0 / 120.0 - h2 * (1.0 / 5040.0 - h2 * (1.0 / 362880.0)))));
        real = 1.0 - h2 * (0.5 - h2 * (1.0 / 24.0 - h2 * (1.0 / 720.0 - h2 * (1.0 / 40320.0 - h2 * (1.0 / 3628800.0)))));
    } else if (th > eps) {
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
    // Jl rho = rho + c1 (phi x rho) + c2 (phi x (phi x rho))
    const double k1[3] = {phi[1] * rho[2] - phi[2] * rho[1], phi[2] * rho[0] - phi[0] * rho[2], phi[0] * rho[1] - phi[1] * rho[0]};
    const double k2[3] = {phi[1] * k1[2] - phi[2] * k1[1], phi[2] * k1[0] - phi[0] * k1[2], phi[0] * k1[1] - phi[1] * k1[0]};
    const double te[3] = {rho[0] + c1 * k1[0] + c2 * k2[0], rho[1] + c1 * k1[1] + c2 * k2[1], rho[2] + c1 * k1[2] + c2 * k2[2]};
    const double qe[4] = {phi[0] * imag, phi[1] * imag, phi[2] * imag, real};
    // t' = te + qe.Act(t);  q' = qe * q
    double rt[3];
    quat_act(qe, P.t, rt);
    const double a[3] = {qe[0], qe[1], qe[2]}, aw = qe[3];
    const double b[3] = {P.q[0], P.q[1], P.q[2]}, bw = P.q[3];
    const double nq[4] = {aw * b[0] + bw * a[0] + (a[1] * b[2] - a[2] * b[1]),
                          aw * b[1] + bw * a[1] + (a[2] * b[0] - a[0] * b[2]),
                          aw * b[2] + bw * a[2] + (a[0] * b[1] - a[1] * b[0]),
                          aw * bw - (a[0] * b[0] + a[1] * b[1] + a[2] * b[2])};
    P.t[0] = te[0] + rt[0]; P.t[1] = te[1] + rt[1]; P.t[2] = te[2] + rt[2];
    P.q[0] = nq[0]; P.q[1] = nq[1]; P.q[2] = nq[2]; P.q[3] = nq[3];
    quat_to_R(P);
}

__device__ __forceinline__ double huber(double x, double delta) {
    const double sx = sqrt(x);
    return (sx < delta) ? x : (2.0 * delta * sx - delta * delta);
}

// torch.linalg.pinv of the symmetric 2x2 [[a, c], [c, b]] (+ optional independent third singular value s3
// of the block-diagonal 3x3) with relative cutoff rcond * sigma_max.
__device__ __forceinline__ void pinv_sym2_blk(double a, double b, double c, double s3, bool has3, double rcond,
                                              double& w00, double& w01, double& w11, double& w22) {
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
__device__ __forceinline__ void inv3(const double* m, double* o) {
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    const double id = 1.0 / det;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// index of (j, k), j <= k, in the packed upper triangle of a 6x6
__device__ __forceinline__ constexpr int tri(int j, int k) { return j * 6 - (j * (j - 1)) / 2 + (k - j); }

struct Geometry {
    double fx, fy, cx, cy, blfx;
};


// Everything a point contributes, gathered once (fp32 buffers widened to fp64 exactly as the reference's
// `.to(torch.double)` does, Optimizer.py:84-85).
template <int GT>
struct PointData {
    bool valid;
    double pw[3];      // pos_Tw
    double obs[3];     // REPROJ/DISP: (u, v, disparity) ; ICP: points_Tc (pixel2point_NED evaluated in fp32)
    double W[3][3];    // REPROJ/DISP: pinv(Sigma_i) (constant); ICP: unused
    double So[9], Sp[9];  // ICP only: obs2_covTc, cov_Tw
};

template <int GT>
__device__ __forceinline__ void load_point(const PgoArgs& a, const Geometry& g, const mvLMParams& lm, int i, bool in_range,
                                           PointData<GT>& d) {
    d.valid = in_range && (a.valid ? (a.valid[i] != 0) : true);
    if (!d.valid) return;
    d.pw[0] = (double)a.pos_Tw[3 * i]; d.pw[1] = (double)a.pos_Tw[3 * i + 1]; d.pw[2] = (double)a.pos_Tw[3 * i + 2];
    if (GT == MV_GRAPH_ICP) {
        // points_Tc = pixel2point_NED(pixel2_uv, pixel2_d, K) built in fp32 (Graphs.py:49-51), then cast
        const float u = a.pixel2_uv[2 * i], v = a.pixel2_uv[2 * i + 1], dd = a.pixel2_d[i];
        d.obs[0] = (double)dd;
        d.obs[1] = (double)(((u - (float)g.cx) * dd) / (float)g.fx);
        d.obs[2] = (double)(((v - (float)g.cy) * dd) / (float)g.fy);
#pragma unroll
        for (int k = 0; k < 9; ++k) { d.So[k] = a.obs2_covTc[9 * (size_t)i + k]; d.Sp[k] = a.cov_Tw[9 * (size_t)i + k]; }
    } else {
        d.obs[0] = (double)a.pixel2_uv[2 * i]; d.obs[1] = (double)a.pixel2_uv[2 * i + 1];
        const double suu = (double)a.pixel2_uv_cov[3 * i], svv = (double)a.pixel2_uv_cov[3 * i + 1],
                     suv = (double)a.pixel2_uv_cov[3 * i + 2];
        double w00, w01, w11, w22;
        if (GT == MV_GRAPH_DISP) {
            d.obs[2] = (double)a.pixel2_disp[i];
            pinv_sym2_blk(suu, svv, suv, (double)a.pixel2_disp_cov[i], true, lm.pinv_rcond, w00, w01, w11, w22);
        } else {
            d.obs[2] = 0.0;
            pinv_sym2_blk(suu, svv, suv, 0.0, false, lm.pinv_rcond, w00, w01, w11, w22);
        }
        d.W[0][0] = w00; d.W[0][1] = w01; d.W[1][0] = w01; d.W[1][1] = w11;
        d.W[0][2] = d.W[2][0] = d.W[1][2] = d.W[2][1] = 0.0;
        d.W[2][2] = w22;
    }
}

// residual block under pose P; returns |r|^2.  pc = T*p_c (ICP) or p_c = T^-1 p_w (REPROJ/DISP).
template <int GT>
__device__ __forceinline__ double residual(const Geometry& g, const Pose& P, const PointData<GT>& d, double* r, double* pc) {
    if (GT == MV_GRAPH_ICP) {
        double rp[3];
        quat_act(P.q, d.obs, rp);
        pc[0] = rp[0] + P.t[0]; pc[1] = rp[1] + P.t[1]; pc[2] = rp[2] + P.t[2];
        r[0] = pc[0] - d.pw[0]; r[1] = pc[1] - d.pw[1]; r[2] = pc[2] - d.pw[2];
        return r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    } else {
        // p_c = T^-1 p_w : Inv = (-q^-1.Act(t), q^-1), Act = q^-1.Act(p_w) + t_inv
        const double qi[4] = {-P.q[0], -P.q[1], -P.q[2], P.q[3]};
        double ti[3], rp[3];
        quat_act(qi, P.t, ti);
        quat_act(qi, d.pw, rp);
        pc[0] = rp[0] - ti[0]; pc[1] = rp[1] - ti[1]; pc[2] = rp[2] - ti[2];
        // point2pixel_NED = homo2cart(p_EDN K^T): u = (fx Y + cx X) / X, v = (fy Z + cy X) / X
        const double X = pc[0];
        double den = fmax(fabs(X), 2.2250738585072014e-308);
        den = (X >= 0.0) ? den : -den;
        r[0] = (g.fx * pc[1] + g.cx * X) / den - d.obs[0];
        r[1] = (g.fy * pc[2] + g.cy * X) / den - d.obs[1];
        double n2 = r[0] * r[0] + r[1] * r[1];
        if (GT == MV_GRAPH_DISP) {
            r[2] = (1.0 / X) * g.blfx - d.obs[2];
            n2 += r[2] * r[2];
        }
        return n2;
    }
}

// One point's contribution to {A_w (21), g_w (6), A_u (21), g_u (6), loss (1)} = acc[55]
template <int GT>
__device__ __forceinline__ void accumulate_point(const Geometry& g, const mvLMParams& lm, const Pose& P,
                                                 const PointData<GT>& d, double (&acc)[NRED]) {
    constexpr int NR = (GT == MV_GRAPH_REPROJ) ? 2 : 3;
    double r[3] = {0, 0, 0}, pc[3];
    const double n2 = residual<GT>(g, P, d, r, pc);
    acc[54] += huber(n2, lm.huber_delta);
    // FastTriggs: s = sqrt(rho'(|r|^2)); both R and J are scaled by s => s^2 on every product
    const double sn = sqrt(n2);
    const double s2 = (sn < lm.huber_delta) ? 1.0 : (lm.huber_delta / sn);

    double J[NR][6];
    double W[NR][NR];
    if (GT == MV_GRAPH_ICP) {
        // J = [I, -skew(T p_c)]
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int c = 0; c < 6; ++c) J[rr][c] = 0.0;
        J[0][0] = J[1][1] = J[2][2] = 1.0;
        J[0][4] = pc[2];  J[0][5] = -pc[1];
        J[1][3] = -pc[2]; J[1][5] = pc[0];
        J[2][3] = pc[1];  J[2][4] = -pc[0];
        // Sigma_i = R Sigma_obs R^T + Sigma_pt ; W_i = pinv(Sigma_i)   (Graphs.py:62-68, Optimizer.py:96-98)
        double T1[9], S[9], Wi[9];
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int y = 0; y < 3; ++y)
                T1[3 * x + y] = P.R[3 * x] * d.So[y] + P.R[3 * x + 1] * d.So[3 + y] + P.R[3 * x + 2] * d.So[6 + y];
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int y = 0; y < 3; ++y)
                S[3 * x + y] = (T1[3 * x] * P.R[3 * y] + T1[3 * x + 1] * P.R[3 * y + 1] + T1[3 * x + 2] * P.R[3 * y + 2]) + d.Sp[3 * x + y];
        inv3(S, Wi);
#pragma unroll
        for (int x = 0; x < NR; ++x)
#pragma unroll
            for (int y = 0; y < NR; ++y) W[x][y] = Wi[3 * x + y];
    } else {
        // G = d p_c / d delta = [-R^T, R^T skew(p_w)]   (3 x 6)
        double G[3][6];
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            const double rt0 = P.R[x], rt1 = P.R[3 + x], rt2 = P.R[6 + x];  // row x of R^T
            G[x][0] = -rt0; G[x][1] = -rt1; G[x][2] = -rt2;
            // R^T skew(p): col0 = R^T (0, pz, -py), col1 = R^T (-pz, 0, px), col2 = R^T (py, -px, 0)
            G[x][3] = rt1 * d.pw[2] - rt2 * d.pw[1];
            G[x][4] = -rt0 * d.pw[2] + rt2 * d.pw[0];
            G[x][5] = rt0 * d.pw[1] - rt1 * d.pw[0];
        }
        const double X = pc[0], Y = pc[1], Z = pc[2], X2 = X * X;
        const double j00 = -g.fx * Y / X2, j01 = g.fx / X, j10 = -g.fy * Z / X2, j12 = g.fy / X;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            J[0][c] = j00 * G[0][c] + j01 * G[1][c];
            J[1][c] = j10 * G[0][c] + j12 * G[2][c];
        }
        if (GT == MV_GRAPH_DISP) {
            const double jd = -g.blfx / X2;
#pragma unroll
            for (int c = 0; c < 6; ++c) J[NR - 1][c] = jd * G[0][c];
        }
#pragma unroll
        for (int x = 0; x < NR; ++x)
#pragma unroll
            for (int y = 0; y < NR; ++y) W[x][y] = d.W[x][y];
    }
    // reference: J_T = J^T @ weight ; A = J_T @ J ; b = -J_T @ R   (PyposeOptimizers.py:170-176)
    double WJ[NR][6], Wr[NR];
#pragma unroll
    for (int x = 0; x < NR; ++x) {
        double t = 0.0;
#pragma unroll
        for (int y = 0; y < NR; ++y) t += W[x][y] * r[y];
        Wr[x] = t;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double tj = 0.0;
#pragma unroll
            for (int y = 0; y < NR; ++y) tj += W[x][y] * J[y][c];
            WJ[x][c] = tj;
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double gwj = 0.0, guj = 0.0;
#pragma unroll
        for (int x = 0; x < NR; ++x) { gwj += J[x][j] * Wr[x]; guj += J[x][j] * r[x]; }
        acc[21 + j] += s2 * gwj;
        acc[48 + j] += s2 * guj;
#pragma unroll
        for (int k = j; k < 6; ++k) {
            double aw = 0.0, au = 0.0;
#pragma unroll
            for (int x = 0; x < NR; ++x) { aw += J[x][j] * WJ[x][k]; au += J[x][j] * J[x][k]; }
            acc[tri(j, k)] += s2 * aw;
            acc[27 + tri(j, k)] += s2 * au;
        }
    }
}

// solve A D = b, b = -gw, by Cholesky (A = L L^T) with the diagonal of A taken from `dg6` (the damped one); every thread solves
// redundantly (uniform control flow).  The two substitutions multiply by the reciprocal pivots the factorisation already has: an
// fp64 division is a ~12 instruction dependent chain, and with up to 17 solves in a rejected step this serial piece was 3.5 k
// cycles each.  Returns false where PyPose reports "Linear solver failed".
__device__ __forceinline__ bool chol_solve6(const double* __restrict__ Aw, const double (&dg6)[6], const double* __restrict__ gw,
                                            double (&D)[6]) {
    double L[6][6], linv[6];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double dd = dg6[j];
#pragma unroll
        for (int k = 0; k < j; ++k) dd -= L[j][k] * L[j][k];
        ok = ok && (dd > 0.0) && (dd < INFINITY);
#ifdef MV_PGO_SQRT_DIV
        const double ljj = sqrt(dd);
        const double inv = 1.0 / ljj;
#else
        // one reciprocal square root per pivot instead of a square root AND a division (each a ~15-instruction dependent fp64
        // sequence on the critical path of every trial); l_jj = dd / sqrt(dd) to ~1 ulp
        const double inv = rsqrt(dd);
        const double ljj = dd * inv;
#endif
        L[j][j] = ljj;
        linv[j] = inv;
#pragma unroll
        for (int i2 = j + 1; i2 < 6; ++i2) {
            double sacc = Aw[tri(j, i2)];
#pragma unroll
            for (int k = 0; k < j; ++k) sacc -= L[i2][k] * L[j][k];
            L[i2][j] = sacc * inv;
        }
    }
    if (!ok) return false;
    double yv[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double sacc = -gw[j];
#pragma unroll
        for (int k = 0; k < j; ++k) sacc -= L[j][k] * yv[k];
        yv[j] = sacc * linv[j];
    }
#pragma unroll
    for (int j = 5; j >= 0; --j) {
        double sacc = yv[j];
#pragma unroll
        for (int k = j + 1; k < 6; ++k) sacc -= L[k][j] * D[k];
        D[j] = sacc * linv[j];
    }
    return true;
}

// TrustRegion.update: quality = (last - loss) / -((J D)^T (2 R + J D)) on the corrected, unweighted J, R
__device__ __forceinline__ double tr_quality(const double (&D)[6], const double* __restrict__ gu, const double* __restrict__ Au,
                                             double last, double loss) {
    double dAd = 0.0, dg = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        dg += D[j] * gu[j];
#pragma unroll
        for (int k = 0; k < 6; ++k) dAd += D[j] * D[k] * Au[(j <= k) ? tri(j, k) : tri(k, j)];
    }
    return (last - loss) / -(2.0 * dg + dAd);
}

// ... and the radius / damping update it drives; returns the branch taken (1: quality > high, 2: > low, 3: shrink)
__device__ __forceinline__ void tr_apply(const mvLMParams& lm, int branch, double& damping, double& tr_down) {
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
__device__ __forceinline__ int tr_update(const mvLMParams& lm, double quality, double& damping, double& tr_down) {
    const int branch = (quality > lm.tr_high) ? 1 : (quality > lm.tr_low) ? 2 : 3;
    tr_apply(lm, branch, damping, tr_down);
    return branch;
}

template <int GT, int NW>
__global__ __launch_bounds__(64 * NW) void pgo_solve_kernel_v1(PgoArgs a, mvLMParams lm) {
    constexpr int PGO_THREADS = 64 * NW;
    __shared__ double red_tab[NW][NRED];
    __shared__ __attribute__((aligned(16))) double red_part[NW == 4 ? NRED * NW * 8 : 1];
    __shared__ __attribute__((aligned(16))) double red_fin[NW == 4 ? NRED + 1 : 1];
    // speculative reject rounds (NW == 4, one point per thread): every point's position / observation for the trial-loss passes, and
    // what each wave found for its trial
    constexpr int SPEC = (NW == 4) ? 1 : 0;
    __shared__ double pt_tab[SPEC ? 6 : 1][SPEC ? 64 * NW : 1];
    __shared__ int pt_valid[SPEC ? 64 * NW : 1];
    __shared__ double spec_res[SPEC ? NW : 1][10];   // per wave: ok, loss, quality, pose t[3] q[4]
    const int prob = blockIdx.x;
    const int tid = threadIdx.x;
    const int beg = a.offsets[prob], end = a.offsets[prob + 1];
    const int npts = end - beg;
    const bool cached = npts <= PGO_THREADS;  // every thread owns (at most) one point for the whole solve

    Geometry g;
    g.fx = (double)a.intrinsics[4 * prob]; g.fy = (double)a.intrinsics[4 * prob + 1];
    g.cx = (double)a.intrinsics[4 * prob + 2]; g.cy = (double)a.intrinsics[4 * prob + 3];
    g.blfx = g.fx * (double)a.baseline[prob];  // K[0,0] * bl in fp64 of the fp32 buffers

    Pose P;
#pragma unroll
    for (int k = 0; k < 3; ++k) P.t[k] = (double)a.init_pose[7 * prob + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) P.q[k] = (double)a.init_pose[7 * prob + 3 + k];
    quat_to_R(P);

    PointData<GT> mine;
    mine.valid = false;
    if (cached) load_point<GT>(a, g, lm, beg + tid, tid < npts, mine);
    if (SPEC && cached) {   // (read behind the barriers of the observation count below)
        pt_valid[tid] = mine.valid ? 1 : 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) { pt_tab[k][tid] = mine.valid ? mine.pw[k] : 0.0; pt_tab[3 + k][tid] = mine.valid ? mine.obs[k] : 0.0; }
    }

    double damping = 1.0 / lm.radius, tr_down = lm.tr_down;
    double loss = 0.0, last = 0.0, loss0 = 0.0;
    bool have_loss = false;
    int steps = 0, patience_count = 0, reject_count = 0;
    bool continual = true;
    int dbg_rounds = 0, dbg_trials = 0;
    int pred_branch = 3;   // trust-region branch the last rejected trial took: the prediction for the following ones

    // Odometry/MACVO.py:303-307: fewer than min_num_point observations => no optimisation, pose stays at the prior
    {
        double nv[1] = {0.0};
        if (cached) {
            nv[0] = mine.valid ? 1.0 : 0.0;
        } else {
            for (int i = beg + tid; i < end; i += PGO_THREADS) nv[0] += (a.valid ? (a.valid[i] != 0) : 1) ? 1.0 : 0.0;
        }
        block_sum<1, NW>(nv, red_tab);
        if ((int)nv[0] < a.min_points) continual = false;
    }

    while (continual) {
        // ------------------------------------------------------------------ build pass
        double acc[NRED];
#pragma unroll
        for (int k = 0; k < NRED; ++k) acc[k] = 0.0;
        if (cached) {
            if (mine.valid) accumulate_point<GT>(g, lm, P, mine, acc);
        } else {
            for (int i = beg + tid; i < end; i += PGO_THREADS) {
                PointData<GT> d;
                load_point<GT>(a, g, lm, i, true, d);
                if (d.valid) accumulate_point<GT>(g, lm, P, d, acc);
            }
        }
        if constexpr (NW == 4) block_sum_build<NW>(acc, red_part, red_fin);
        else block_sum<NRED, NW>(acc, red_tab);
        double* Aw = acc;
        const double* gw = acc + 21;
        const double* Au = acc + 27;
        const double* gu = acc + 48;

        if (!have_loss) { loss = acc[54]; loss0 = acc[54]; have_loss = true; }
        last = loss;
        reject_count = 0;

        // A.diagonal().clamp_(min, max)
#pragma unroll
        for (int j = 0; j < 6; ++j) Aw[tri(j, j)] = fmin(fmax(Aw[tri(j, j)], lm.diag_min), lm.diag_max);

        // ------------------------------------------------------------------ inner damping / reject loop
        // Round 3: once a step's first trial has been rejected, the following trials are evaluated FOUR AT A TIME, one per wave.
        // A rejected trial leaves the pose where it was, multiplies the damping into A's diagonal once more and updates the trust
        // region through one of the three branches of TrustRegion.update — in practice the SAME branch trial after trial (at the
        // end of a solve the unweighted model the quality is measured against predicts an increase: quality > tr_high although
        // the loss went up) — so the inputs of the next trials are known before the previous ones have been evaluated, PROVIDED
        // those are rejected through the predicted branch (= the branch of the last rejected trial).  Wave w replays that scalar
        // recurrence w times, solves its own system, moves its own copy of the pose and sums the trial loss over ALL points (four
        // 64-point passes over the LDS point table, each reduced with the same DPP tree and added in the same order as the block
        // reduction of the sequential form: identical bits).  Then every thread walks the four results in order with the sequential
        // form's own update code and stops at the first trial that is accepted, fails to factorise, or was rejected through another
        // branch (the later results of the round are then discarded, the prediction becomes that branch and the next round starts
        // from the true state).  A step that exhausts its 16 rejections costs 1 + 4 rounds instead of 17 sequential trials.
        while (last <= loss) {
            dbg_trials += 1;
            const bool spec_round = SPEC && cached && reject_count >= 1 && a.spec != 0;
            if (spec_round) dbg_rounds += 1;
            if (!spec_round) {
#pragma unroll
                for (int j = 0; j < 6; ++j) Aw[tri(j, j)] += Aw[tri(j, j)] * damping;
                double dg6[6], D[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) dg6[j] = Aw[tri(j, j)];
                if (!chol_solve6(Aw, dg6, gw, D)) break;  // "Linear solver failed. Breaking optimization step..."

                const Pose P_prev = P;
                se3_left_update(P, D);

                // loss at the trial pose (RobustModel.loss: unweighted, uncorrected)
                double la[1] = {0.0};
                if (cached) {
                    if (mine.valid) {
                        double r[3] = {0, 0, 0}, pc[3];
                        la[0] = huber(residual<GT>(g, P, mine, r, pc), lm.huber_delta);
                    }
                } else {
                    for (int i = beg + tid; i < end; i += PGO_THREADS) {
                        PointData<GT> d;
                        load_point<GT>(a, g, lm, i, true, d);
                        if (d.valid) {
                            double r[3] = {0, 0, 0}, pc[3];
                            la[0] += huber(residual<GT>(g, P, d, r, pc), lm.huber_delta);
                        }
                    }
                }
                block_sum<1, NW>(la, red_tab);
                loss = la[0];

                const double quality = tr_quality(D, gu, Au, last, loss);
                pred_branch = tr_update(lm, quality, damping, tr_down);

                if (last < loss && reject_count < lm.reject) {  // reject step
                    P = P_prev;
                    loss = last;
                    reject_count += 1;
                } else {
                    break;
                }
            } else {
                const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
                // ---- this wave's trial: the (wv + 1)-th from here, assuming the wv before it are rejected through the predicted branch
                double dg6[6], damp_s = damping, trd_s = tr_down;
#pragma unroll
                for (int j = 0; j < 6; ++j) dg6[j] = Aw[tri(j, j)];
                for (int i = 0; i <= wv; ++i) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) dg6[j] += dg6[j] * damp_s;
                    if (i < wv) tr_apply(lm, pred_branch, damp_s, trd_s);
                }
                double D[6];
                const bool ok = chol_solve6(Aw, dg6, gw, D);
                Pose Pw = P;
                double loss_w = 0.0, quality_w = 0.0;
                if (ok) {
                    se3_left_update(Pw, D);
                    double part[NW];
#pragma unroll
                    for (int c = 0; c < NW; ++c) {
                        const int i = c * 64 + lane;
                        double v = 0.0;
                        if (pt_valid[i]) {
                            PointData<GT> d;
                            d.valid = true;
#pragma unroll
                            for (int k = 0; k < 3; ++k) { d.pw[k] = pt_tab[k][i]; d.obs[k] = pt_tab[3 + k][i]; }
                            double r[3] = {0, 0, 0}, pc[3];
                            v = huber(residual<GT>(g, Pw, d, r, pc), lm.huber_delta);
                        }
                        part[c] = wave_sum_dpp(v);
                    }
                    loss_w = part[0];
#pragma unroll
                    for (int c = 1; c < NW; ++c) loss_w += part[c];
                    quality_w = tr_quality(D, gu, Au, last, loss_w);
                }
                if (lane == 0) {
                    double* o = spec_res[wv];
                    o[0] = ok ? 1.0 : 0.0; o[1] = loss_w; o[2] = quality_w;
                    o[3] = Pw.t[0]; o[4] = Pw.t[1]; o[5] = Pw.t[2];
                    o[6] = Pw.q[0]; o[7] = Pw.q[1]; o[8] = Pw.q[2]; o[9] = Pw.q[3];
                }
                __syncthreads();
                // ---- the sequential form's bookkeeping over the four results
                bool leave = false;
                for (int i = 0; i < NW; ++i) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) Aw[tri(j, j)] += Aw[tri(j, j)] * damping;
                    const double* o = spec_res[i];
                    if (o[0] == 0.0) { leave = true; break; }       // "Linear solver failed"
                    loss = o[1];
                    const int branch = tr_update(lm, o[2], damping, tr_down);
                    if (last < loss && reject_count < lm.reject) {   // reject step
                        loss = last;
                        reject_count += 1;
                        if (branch != pred_branch) { pred_branch = branch; break; }   // the later trials of this round started from other inputs
                    } else {
                        P.t[0] = o[3]; P.t[1] = o[4]; P.t[2] = o[5];
                        P.q[0] = o[6]; P.q[1] = o[7]; P.q[2] = o[8]; P.q[3] = o[9];
                        quat_to_R(P);
                        leave = true;
                        break;
                    }
                }
                __syncthreads();   // spec_res is rewritten by the next round
                if (leave) break;
            }
        }

        // ------------------------------------------------------------------ StopOnPlateau.step(loss)
        steps += 1;
        if (steps >= lm.max_steps) continual = false;
        if ((last - loss) < lm.decreasing) patience_count += 1; else patience_count = 0;
        if (patience_count >= lm.patience) continual = false;
        if (lm.stop_on_reject > 0 && reject_count >= lm.stop_on_reject) continual = false;
    }

    if (tid == 0) {
        double* o = a.out_pose + 7 * (size_t)prob;
        o[0] = P.t[0]; o[1] = P.t[1]; o[2] = P.t[2];
        o[3] = P.q[0]; o[4] = P.q[1]; o[5] = P.q[2]; o[6] = P.q[3];
        double* inf = a.out_info + 4 * (size_t)prob;
        inf[0] = loss; inf[1] = (double)steps; inf[2] = (double)reject_count; inf[3] = loss0;
        if (a.spec == 2) inf[3] = (double)(dbg_rounds * 1000 + dbg_trials);
        if (a.out_pose_f32) {
            float* of = a.out_pose_f32 + 7 * (size_t)prob;  // write_graph_data: pose = motion.float()
            of[0] = (float)P.t[0]; of[1] = (float)P.t[1]; of[2] = (float)P.t[2];
            of[3] = (float)P.q[0]; of[4] = (float)P.q[1]; of[5] = (float)P.q[2]; of[6] = (float)P.q[3];
        }
    }
}

}  // namespace

// round-3 form of the solve kept for one A/B (MV_PGO_V1=1); C++ linkage: not part of the C ABI
int mv_pgo_solve_v1_cxx(int nprob, const int32_t* offsets, int graph_type, const float* init_pose,
                            const float* intrinsics, const float* baseline, const float* pos_Tw, const double* cov_Tw,
                            const float* pixel2_uv, const float* pixel2_d, const float* pixel2_disp,
                            const float* pixel2_disp_cov, const float* pixel2_uv_cov, const double* obs2_covTc,
                            const uint8_t* valid, int min_points, const mvLMParams* params, double* out_pose,
                            double* out_info, float* out_pose_f32, mvStream_t stream) {
    MV_CHECK_ARG(nprob >= 0 && params);
    if (nprob == 0) return MV_OK;
    MV_CHECK_ARG(offsets && init_pose && intrinsics && baseline && pos_Tw && pixel2_uv && out_pose && out_info);
    MV_CHECK_ARG(params->max_steps >= 1 && params->reject >= 0 && params->stop_on_reject >= 0 && params->radius > 0 && params->huber_delta > 0);
    PgoArgs a{offsets, init_pose, intrinsics, baseline, pos_Tw, cov_Tw, pixel2_uv, pixel2_d, pixel2_disp,
              pixel2_disp_cov, pixel2_uv_cov, obs2_covTc, valid, min_points, out_pose, out_info, out_pose_f32, 1};
    {
        static int spec = -1;   // MV_PGO_SPEC=0: every trial of the reject loop sequentially (A/B knob)
        if (spec < 0) { const char* e = getenv("MV_PGO_SPEC"); spec = e ? atoi(e) : 1; }
        a.spec = spec;
    }
    hipStream_t s = (hipStream_t)stream;
    // latency variant (4 waves per problem, one point per thread in registers) for small batches; throughput variant
    // (1 wave per problem, 4x more problems resident per CU) once the batch alone fills the chip
    const bool wide = nprob < 512;
    dim3 grid(nprob), block(wide ? 256 : 64);
#define MV_PGO(G)                                                                             \
    if (wide) hipLaunchKernelGGL((pgo_solve_kernel_v1<G, 4>), grid, block, 0, s, a, *params);    \
    else hipLaunchKernelGGL((pgo_solve_kernel_v1<G, 1>), grid, block, 0, s, a, *params)
    switch (graph_type) {
        case MV_GRAPH_ICP:
            MV_CHECK_ARG(cov_Tw && obs2_covTc && pixel2_d);
            MV_PGO(MV_GRAPH_ICP);
            break;
        case MV_GRAPH_REPROJ:
            MV_CHECK_ARG(pixel2_uv_cov);
            MV_PGO(MV_GRAPH_REPROJ);
            break;
        case MV_GRAPH_DISP:
            MV_CHECK_ARG(pixel2_uv_cov && pixel2_disp && pixel2_disp_cov);
            MV_PGO(MV_GRAPH_DISP);
            break;
        default:
            return MV_ERR_INVALID_ARG;
    }
#undef MV_PGO
    return mv_launch_status();
}
