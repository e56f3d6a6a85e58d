// SURVEY §8(f) rank 4 — device-resident VisualMap: per-frame registration, output poses, MotionInterpolate
//
// Replaces, for the tracking map of one sequence:
//   Module/Map/VisualMap.py:15-133 (SoA stores frames / points / match + six edge tables) and Module/Map/Graph.py:19-298
//   (TensorBundle / AutoScalingBundle.push :86-104, DenseEdge_Multi.add :183-186, SparseEdge_Multi.add :144-147,
//   SingleEdge.set :228-229) as they are driven by Odometry/MACVO.py:158-171 (initialize), :244-311 (run_pair: MatchObs.init,
//   `match_obs[mask]`, points.push(...[mask]), push_keyframe, the six edge updates, the lost-track flag) — the reference does
//   all of it on the CPU after ~25 `.cpu()` copies per frame (MACVO.py:235-266);
//   Odometry/Interface.py:47-49 (body poses T_BS @ pose @ T_BS^-1 written to poses.npy);
//   Module/MapProcessor.py:52-76 (MotionInterpolate.elaborate_map) + Utility/Math.py:96-133 (interpolate_pose, NormalizeQuat).
//
// gfx950 design: all of this is a few KB per frame — one workgroup per call.  mv_map_append compacts the frame's kept rows
// (valid mask of mv_obs_filter, order preserved: `bundle[mask]`) with a wave-ballot scan and scatters them into the stores at
// the device-side running counts, so a frame is registered with ZERO host synchronisation and zero D2H copies; the host only
// tracks capacity UPPER bounds (rows pushed <= rows selected).
#include "common.h"
#include <math.h>

namespace {

struct Se3d {
    double t[3], q[4];
};

__device__ __forceinline__ void q_act(const double* q, const double* p, double* o) {   // PyPose SO3_Act
    double u0 = q[1] * p[2] - q[2] * p[1], u1 = q[2] * p[0] - q[0] * p[2], u2 = q[0] * p[1] - q[1] * p[0];
    u0 += u0; u1 += u1; u2 += u2;
    o[0] = p[0] + q[3] * u0 + (q[1] * u2 - q[2] * u1);
    o[1] = p[1] + q[3] * u1 + (q[2] * u0 - q[0] * u2);
    o[2] = p[2] + q[3] * u2 + (q[0] * u1 - q[1] * u0);
}
__device__ __forceinline__ void q_mul(const double* a, const double* b, double* o) {   // PyPose SO3_Mul (x, y, z, w)
    o[0] = a[3] * b[0] + b[3] * a[0] + (a[1] * b[2] - a[2] * b[1]);
    o[1] = a[3] * b[1] + b[3] * a[1] + (a[2] * b[0] - a[0] * b[2]);
    o[2] = a[3] * b[2] + b[3] * a[2] + (a[0] * b[1] - a[1] * b[0]);
    o[3] = a[3] * b[3] - (a[0] * b[0] + a[1] * b[1] + a[2] * b[2]);
}
__device__ __forceinline__ Se3d se3_mul(const Se3d& a, const Se3d& b) {
    Se3d o;
    double r[3];
    q_act(a.q, b.t, r);
    o.t[0] = a.t[0] + r[0]; o.t[1] = a.t[1] + r[1]; o.t[2] = a.t[2] + r[2];
    q_mul(a.q, b.q, o.q);
    return o;
}
__device__ __forceinline__ Se3d se3_inv(const Se3d& a) {
    Se3d o;
    o.q[0] = -a.q[0]; o.q[1] = -a.q[1]; o.q[2] = -a.q[2]; o.q[3] = a.q[3];
    double r[3];
    q_act(o.q, a.t, r);
    o.t[0] = -r[0]; o.t[1] = -r[1]; o.t[2] = -r[2];
    return o;
}
__device__ __forceinline__ Se3d se3_normq(Se3d a) {   // Utility/Math.py:124-133 NormalizeQuat
    const double n = sqrt(a.q[0] * a.q[0] + a.q[1] * a.q[1] + a.q[2] * a.q[2] + a.q[3] * a.q[3]);
    a.q[0] /= n; a.q[1] /= n; a.q[2] /= n; a.q[3] /= n;
    return a;
}
__device__ __forceinline__ void skew_sq_apply(const double* k, const double* v, double c1, double c2, double* o) {
    // o = v + c1 * (k x v) + c2 * (k x (k x v))
    const double a0 = k[1] * v[2] - k[2] * v[1], a1 = k[2] * v[0] - k[0] * v[2], a2 = k[0] * v[1] - k[1] * v[0];
    const double b0 = k[1] * a2 - k[2] * a1, b1 = k[2] * a0 - k[0] * a2, b2 = k[0] * a1 - k[1] * a0;
    o[0] = v[0] + c1 * a0 + c2 * b0;
    o[1] = v[1] + c1 * a1 + c2 * b1;
    o[2] = v[2] + c1 * a2 + c2 * b2;
}
// PyPose SE3_Log: phi = SO3_Log(q), rho = Jl^-1(phi) t   (tangent = [rho, phi])
__device__ __forceinline__ void se3_log(const Se3d& a, double* xi) {
    const double eps = 2.220446049250313e-16;
    double v[3] = {a.q[0], a.q[1], a.q[2]}, w = a.q[3];
    const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    // PyPose SO3_Log: factor = 2 atan(|v| / w) / |v| (w != 0), with the small-|v| series 2/w - 2 |v|^2 / (3 w^3)
    double factor;
    if (n > eps) {
        factor = (fabs(w) > eps) ? 2.0 * atan(n / w) / n : (w >= 0 ? M_PI : -M_PI) / n;
    } else {
        factor = 2.0 / w - 2.0 * n * n / (3.0 * w * w * w);
    }
    double phi[3] = {v[0] * factor, v[1] * factor, v[2] * factor};
    const double th = sqrt(phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2]);
    // so3_Jl_inv = I - K/2 + c K^2, c = (1 - th cos(th/2) / (2 sin(th/2))) / th^2, series 1/12 + th^2/720
    double c;
    if (th > eps) {
        const double h = 0.5 * th;
        c = (1.0 - th * cos(h) / (2.0 * sin(h))) / (th * th);
    } else {
        c = 1.0 / 12.0 + th * th / 720.0;
    }
    skew_sq_apply(phi, a.t, -0.5, c, xi);
    xi[3] = phi[0]; xi[4] = phi[1]; xi[5] = phi[2];
}
// PyPose se3_Exp: t = Jl(phi) rho, q = so3_Exp(phi)
__device__ __forceinline__ Se3d se3_exp(const double* xi) {
    const double eps = 2.220446049250313e-16;
    const double* phi = xi + 3;
    const double th = sqrt(phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2]);
    const double th2 = th * th, th4 = th2 * th2;
    Se3d o;
    double c1, c2, imag, real;
    if (th > eps) {
        c1 = (1.0 - cos(th)) / th2;
        c2 = (th - sin(th)) / (th2 * th);
        imag = sin(0.5 * th) / th;
        real = cos(0.5 * th);
    } else {
        c1 = 0.5 - th2 / 24.0;
        c2 = 1.0 / 6.0 - th2 / 120.0;
        imag = 0.5 - th2 / 48.0 + th4 / 3840.0;
        real = 1.0 - th2 / 8.0 + th4 / 384.0;
    }
    skew_sq_apply(phi, xi, c1, c2, o.t);
    o.q[0] = phi[0] * imag; o.q[1] = phi[1] * imag; o.q[2] = phi[2] * imag; o.q[3] = real;
    return o;
}
__device__ __forceinline__ Se3d load_pose(const float* p) {
    Se3d o;
    o.t[0] = p[0]; o.t[1] = p[1]; o.t[2] = p[2];
    o.q[0] = p[3]; o.q[1] = p[4]; o.q[2] = p[5]; o.q[3] = p[6];
    return o;
}

// ------------------------------------------------------------------------------------------------ registration
__global__ __launch_bounds__(256) void map_append_kernel(mvMapFrame fr, mvMapStores st) {
    __shared__ int wave_cnt[4];
    __shared__ int base_match, base_point, frame_idx;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int64_t* cnt = st.counts;   // {n_frames, n_match, n_points, n_frames_need_interp}
    if (t == 0) {
        frame_idx = (int)cnt[0];
        base_match = (int)cnt[1];
        base_point = (int)cnt[2];
    }
    __syncthreads();
    const int F = frame_idx, M0 = base_match, P0 = base_point;
    // row offsets are device-side state: a caller whose capacity bookkeeping slipped must get an error, not a scribble
    if ((int64_t)F >= st.cap_frames || (int64_t)M0 + fr.n_rows > st.cap_match || (int64_t)P0 + fr.n_rows > st.cap_points) {
        if (t == 0) {
            cnt[4] += 1;
            if (fr.out_frame_idx) fr.out_frame_idx[0] = -1;
        }
        return;
    }
    int kept_total = 0;
    // rows are processed in chunks of 256 so that any n_rows works; order is preserved (== `bundle[mask]`)
    for (int r0 = 0; r0 < fr.n_rows; r0 += 256) {
        const int r = r0 + t;
        const bool keep = r < fr.n_rows && (fr.valid ? fr.valid[r] != 0 : true);
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int before = kept_total;
        for (int w = 0; w < wave; ++w) before += wave_cnt[w];
        const int chunk = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        if (keep) {
            const int k = before + __popcll(bal & ((1ull << lane) - 1ull));
            const size_t m = (size_t)M0 + k, p = (size_t)P0 + k, N = (size_t)fr.table_stride;
            // MatchObs (VisualMap.py:51-69; sources: Odometry/MACVO.py:244-266)
            st.pixel1_uv[2 * m] = fr.kp0[2 * r]; st.pixel1_uv[2 * m + 1] = fr.kp0[2 * r + 1];
            st.pixel2_uv[2 * m] = fr.kp1[2 * r]; st.pixel2_uv[2 * m + 1] = fr.kp1[2 * r + 1];
            st.pixel1_d[m] = fr.vals[0 * N + r];
            st.pixel1_disp[m] = fr.vals[1 * N + r];
            st.pixel1_disp_cov[m] = fr.vals[2 * N + r];
            st.pixel1_d_cov[m] = fr.vals[3 * N + r];
            st.pixel2_d[m] = fr.vals[4 * N + r];
            st.pixel2_disp[m] = fr.vals[5 * N + r];
            st.pixel2_disp_cov[m] = fr.vals[6 * N + r];
            st.pixel2_d_cov[m] = fr.vals[7 * N + r];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                st.pixel1_uv_cov[3 * m + c] = fr.sigma0[3 * r + c];
                st.pixel2_uv_cov[3 * m + c] = fr.sigma1[3 * r + c];
                st.pos_Tw[3 * p + c] = fr.pos_Tw[3 * r + c];
                st.color[3 * p + c] = fr.color ? fr.color[3 * r + c] : (uint8_t)0;
            }
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                st.obs1_covTc[9 * m + c] = fr.cov0[9 * (size_t)r + c];
                st.obs2_covTc[9 * m + c] = fr.cov1[9 * (size_t)r + c];
                st.cov_Tw[9 * p + c] = fr.cov0_world[9 * (size_t)r + c];
            }
            // edges: point -> match (new point: degree 0 -> slot 0), match -> point / frame1 / frame2
            int64_t* pe = st.point2match_edges + p * st.max_pt_obs;
            pe[0] = (int64_t)m;
            for (int c = 1; c < st.max_pt_obs; ++c) pe[c] = -1;
            st.point2match_deg[p] = 1;
            st.match2point[m] = (int64_t)p;
            st.match2frame1[m] = (int64_t)fr.prev_frame;
            st.match2frame2[m] = (int64_t)F;
        }
        kept_total += chunk;
        __syncthreads();
    }
    if (t == 0) {
        // push_keyframe (MACVO.py:339-347): the new frame enters the map at the motion-model prior; the optimised pose is
        // written over it later (write_graph_data, Optimizer.py:104-108)
        const size_t f = (size_t)F;
#pragma unroll
        for (int c = 0; c < 9; ++c) st.K[9 * f + c] = fr.K[c];
        st.baseline[f] = fr.baseline;
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            st.pose[7 * f + c] = fr.prior_pose ? fr.prior_pose[c] : (c == 6 ? 1.f : 0.f);
            st.T_BS[7 * f + c] = fr.T_BS[c];
        }
        st.time_ns[f] = fr.time_ns;
        // lost track (MACVO.py:303-307): fewer than min_num_point observations -> flagged for interpolation
        const bool lost = fr.prev_frame >= 0 && kept_total < fr.min_num_point;
        st.need_interp[f] = lost ? 1 : 0;
        // new frame's edge rows (AutoScalingBundle.push :97-104): no ranges yet
        for (int c = 0; c < 2 * st.max_frame_range; ++c) {
            st.frame2match_ranges[2 * st.max_frame_range * f + c] = -1;
            st.frame2map_ranges[2 * st.max_frame_range * f + c] = -1;
        }
        st.frame2match_num[f] = 0;
        st.frame2map_num[f] = 0;
        if (fr.prev_frame >= 0) {
            // frame2match.add(prev_frame, M0, n); frame2match.add(frame, M0, n)  (MACVO.py:290-291, Graph.py:183-186)
            const size_t pf = (size_t)fr.prev_frame;
            const int64_t np = st.frame2match_num[pf];
            if (np < st.max_frame_range) {
                st.frame2match_ranges[2 * (st.max_frame_range * pf + np)] = M0;
                st.frame2match_ranges[2 * (st.max_frame_range * pf + np) + 1] = kept_total;
                st.frame2match_num[pf] = np + 1;
            } else {
                cnt[4] += 1;   // DenseEdge_Multi.add raises when a frame runs out of range slots (Graph.py:183-186)
            }
            st.frame2match_ranges[2 * st.max_frame_range * f] = M0;
            st.frame2match_ranges[2 * st.max_frame_range * f + 1] = kept_total;
            st.frame2match_num[f] = 1;
        }
        cnt[0] = F + 1;
        cnt[1] = M0 + kept_total;
        cnt[2] = P0 + kept_total;
        if (lost) cnt[3] += 1;
        if (fr.out_frame_idx) fr.out_frame_idx[0] = F;
    }
}

// dense-mapping tail: map_points.push + frame2map.add for the newest frame (Odometry/MACVO.py:329-337)
__global__ __launch_bounds__(256) void map_append_points_kernel(mvMapStores st, int n, const float* __restrict__ pos, const double* __restrict__ cov,
                                                                 const uint8_t* __restrict__ color) {
    __shared__ int base, frame, ok;
    int64_t* cnt = st.counts;
    if (threadIdx.x == 0) {
        base = (int)cnt[5];
        frame = (int)cnt[0] - 1;
        const bool fits = frame >= 0 && (int64_t)base + n <= st.cap_map_points && st.frame2map_num[frame] < st.max_frame_range;
        ok = fits ? 1 : 0;
        if (!fits) cnt[4] += 1;
    }
    __syncthreads();
    if (!ok) return;
    const size_t P0 = (size_t)base;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            st.mp_pos_Tw[3 * (P0 + i) + c] = pos[3 * (size_t)i + c];
            st.mp_color[3 * (P0 + i) + c] = color ? color[3 * (size_t)i + c] : (uint8_t)0;
        }
#pragma unroll
        for (int c = 0; c < 9; ++c) st.mp_cov_Tw[9 * (P0 + i) + c] = cov[9 * (size_t)i + c];
    }
    if (threadIdx.x == 0) {
        const size_t f = (size_t)frame;
        const int64_t k = st.frame2map_num[f];
        st.frame2map_ranges[2 * (st.max_frame_range * f + k)] = (int64_t)P0;
        st.frame2map_ranges[2 * (st.max_frame_range * f + k) + 1] = (int64_t)n;
        st.frame2map_num[f] = k + 1;
        cnt[5] = (int64_t)P0 + n;
    }
}

// body poses of Odometry/Interface.py:47-49: T_BS @ pose @ T_BS^-1, float32 arithmetic like pp.SE3(float32 tensors)
__global__ void body_poses_kernel(const float* __restrict__ pose, const float* __restrict__ T_BS, int T, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    const float* a = T_BS + 7 * (size_t)i;
    const float* b = pose + 7 * (size_t)i;
    auto act = [](const float* q, const float* p, float* o) {
        float u0 = q[1] * p[2] - q[2] * p[1], u1 = q[2] * p[0] - q[0] * p[2], u2 = q[0] * p[1] - q[1] * p[0];
        u0 += u0; u1 += u1; u2 += u2;
        o[0] = (p[0] + q[3] * u0) + (q[1] * u2 - q[2] * u1);
        o[1] = (p[1] + q[3] * u1) + (q[2] * u0 - q[0] * u2);
        o[2] = (p[2] + q[3] * u2) + (q[0] * u1 - q[1] * u0);
    };
    auto mulq = [](const float* x, const float* y, float* o) {
        o[0] = (x[3] * y[0] + y[3] * x[0]) + (x[1] * y[2] - x[2] * y[1]);
        o[1] = (x[3] * y[1] + y[3] * x[1]) + (x[2] * y[0] - x[0] * y[2]);
        o[2] = (x[3] * y[2] + y[3] * x[2]) + (x[0] * y[1] - x[1] * y[0]);
        o[3] = x[3] * y[3] - ((x[0] * y[0] + x[1] * y[1]) + x[2] * y[2]);
    };
    // ab = T_BS * pose
    float r[3], ab_t[3], ab_q[4];
    act(a + 3, b, r);
    ab_t[0] = a[0] + r[0]; ab_t[1] = a[1] + r[1]; ab_t[2] = a[2] + r[2];
    mulq(a + 3, b + 3, ab_q);
    // inv(T_BS) = (-q^-1 . t, q^-1)
    const float qi[4] = {-a[3], -a[4], -a[5], a[6]};
    float ti[3];
    act(qi, a, ti);
    ti[0] = -ti[0]; ti[1] = -ti[1]; ti[2] = -ti[2];
    float* o = out + 7 * (size_t)i;
    act(ab_q, ti, r);
    o[0] = ab_t[0] + r[0]; o[1] = ab_t[1] + r[1]; o[2] = ab_t[2] + r[2];
    mulq(ab_q, qi, o + 3);
}

// MotionInterpolate.elaborate_map (Module/MapProcessor.py:57-76), one workgroup, float64:
//   motions[i] = pose[i]^-1 pose[i+1]; a motion INTO a need_interp frame (except the first / last two) is replaced by the
//   se3-linear interpolation between its nearest good neighbours (interpolate_pose, Utility/Math.py:96-121); the track is
//   rebuilt as pose[0] @ cumprod(NormalizeQuat(motions)) and cast back to float32.
__global__ __launch_bounds__(256) void motion_interpolate_kernel(float* __restrict__ pose, const uint8_t* __restrict__ need_interp,
                                                                  int T, double* __restrict__ motions /* [T-1, 7] scratch */,
                                                                  int32_t* __restrict__ out_count) {
    const int t = threadIdx.x;
    const int Mn = T - 1;
    for (int i = t; i < Mn; i += blockDim.x) {
        const Se3d m = se3_mul(se3_inv(load_pose(pose + 7 * (size_t)i)), load_pose(pose + 7 * (size_t)(i + 1)));
        double* o = motions + 7 * (size_t)i;
        o[0] = m.t[0]; o[1] = m.t[1]; o[2] = m.t[2]; o[3] = m.q[0]; o[4] = m.q[1]; o[5] = m.q[2]; o[6] = m.q[3];
    }
    __threadfence_block();
    __syncthreads();
    auto bad = [&](int i) { return i >= 2 && i < Mn - 2 && need_interp[i + 1] != 0; };   // bad_mask[:2] = bad_mask[-2:] = False
    auto load_m = [&](int i) {
        const double* o = motions + 7 * (size_t)i;
        Se3d m;
        m.t[0] = o[0]; m.t[1] = o[1]; m.t[2] = o[2]; m.q[0] = o[3]; m.q[1] = o[4]; m.q[2] = o[5]; m.q[3] = o[6];
        return m;
    };
    int n_bad = 0;
    // interpolated motions go to registers first (a bad motion never serves as a neighbour, so order does not matter)
    for (int i = t; i < Mn; i += blockDim.x) {
        if (!bad(i)) continue;
        ++n_bad;
        int s = i - 1, e = i + 1;
        while (bad(s)) --s;
        while (bad(e)) ++e;
        const Se3d Ps = load_m(s), Pe = load_m(e);
        double xi[6];
        se3_log(se3_mul(Pe, se3_inv(Ps)), xi);
        // torch divides two int64 tensors in the default dtype (float32): Utility/Math.py:114
        const double prop = (double)((float)(i - s) / (float)(e - s));
#pragma unroll
        for (int c = 0; c < 6; ++c) xi[c] *= prop;
        const Se3d m = se3_mul(se3_exp(xi), Ps);
        double* o = motions + 7 * (size_t)i;   // safe: neighbours s, e are good entries, never rewritten
        o[0] = m.t[0]; o[1] = m.t[1]; o[2] = m.t[2]; o[3] = m.q[0]; o[4] = m.q[1]; o[5] = m.q[2]; o[6] = m.q[3];
    }
    n_bad = wave_sum(n_bad);
    __shared__ int tot;
    if (t == 0) tot = 0;
    __syncthreads();
    if ((t & 63) == 0 && n_bad) atomicAdd(&tot, n_bad);
    __threadfence_block();
    __syncthreads();
    if (t == 0) {
        if (out_count) out_count[0] = tot;
        // cumulative product with NormalizeQuat on both operands of every product (MapProcessor.py:73), in order
        Se3d acc = load_pose(pose);
        const Se3d p0 = acc;
        Se3d run;
        for (int i = 0; i < Mn; ++i) {
            const Se3d m = load_m(i);
            run = i == 0 ? m : se3_mul(se3_normq(run), se3_normq(m));
            acc = se3_mul(p0, run);
            float* o = pose + 7 * (size_t)(i + 1);
            o[0] = (float)acc.t[0]; o[1] = (float)acc.t[1]; o[2] = (float)acc.t[2];
            o[3] = (float)acc.q[0]; o[4] = (float)acc.q[1]; o[5] = (float)acc.q[2]; o[6] = (float)acc.q[3];
        }
    }
}

}  // namespace

extern "C" int mv_map_append(const mvMapFrame* frame, const mvMapStores* stores, mvStream_t stream) {
    MV_CHECK_ARG(frame && stores);
    const mvMapFrame& f = *frame;
    const mvMapStores& s = *stores;
    MV_CHECK_ARG(f.n_rows >= 0 && f.table_stride >= f.n_rows && f.K && f.T_BS && s.counts);
    MV_CHECK_ARG(f.n_rows == 0 || (f.kp0 && f.kp1 && f.vals && f.sigma0 && f.sigma1 && f.cov0 && f.cov1 && f.pos_Tw && f.cov0_world));
    MV_CHECK_ARG(s.max_pt_obs >= 1 && s.max_frame_range >= 1);
    MV_CHECK_ARG(s.cap_frames > 0 && s.cap_match > 0 && s.cap_points > 0);
    MV_CHECK_ARG(s.K && s.baseline && s.pose && s.T_BS && s.need_interp && s.time_ns && s.pos_Tw && s.cov_Tw && s.color);
    MV_CHECK_ARG(s.pixel1_uv && s.pixel2_uv && s.pixel1_d && s.pixel2_d && s.pixel1_disp && s.pixel2_disp && s.pixel1_disp_cov &&
                 s.pixel2_disp_cov && s.obs1_covTc && s.obs2_covTc && s.pixel1_uv_cov && s.pixel2_uv_cov && s.pixel1_d_cov &&
                 s.pixel2_d_cov);
    MV_CHECK_ARG(s.frame2match_ranges && s.frame2match_num && s.frame2map_ranges && s.frame2map_num && s.match2frame1 &&
                 s.match2frame2 && s.match2point && s.point2match_edges && s.point2match_deg);
    hipLaunchKernelGGL(map_append_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, f, s);
    return mv_launch_status();
}

extern "C" int mv_map_append_points(const mvMapStores* stores, int n, const float* pos_Tw, const double* cov, const uint8_t* color,
                                    mvStream_t stream) {
    MV_CHECK_ARG(stores && n >= 0);
    const mvMapStores& s = *stores;
    MV_CHECK_ARG(s.counts && s.frame2map_ranges && s.frame2map_num && s.max_frame_range >= 1);
    MV_CHECK_ARG(s.cap_map_points > 0 && s.mp_pos_Tw && s.mp_cov_Tw && s.mp_color);
    MV_CHECK_ARG(n == 0 || (pos_Tw && cov));
    hipLaunchKernelGGL(map_append_points_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, s, n, pos_Tw, cov, color);
    return mv_launch_status();
}

extern "C" int mv_body_poses(const float* pose, const float* T_BS, int T, float* out, mvStream_t stream) {
    MV_CHECK_ARG(T >= 0);
    if (T == 0) return MV_OK;
    MV_CHECK_ARG(pose && T_BS && out);
    hipLaunchKernelGGL(body_poses_kernel, dim3(mv_ceil_div(T, 256)), dim3(256), 0, (hipStream_t)stream, pose, T_BS, T, out);
    return mv_launch_status();
}

extern "C" int mv_motion_interpolate(float* pose, const uint8_t* need_interp, int T, double* scratch, int32_t* out_count,
                                     mvStream_t stream) {
    MV_CHECK_ARG(T >= 0);
    if (T < 2) return MV_OK;
    MV_CHECK_ARG(pose && need_interp && scratch);
    hipLaunchKernelGGL(motion_interpolate_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pose, need_interp, T, scratch,
                       out_count);
    return mv_launch_status();
}
