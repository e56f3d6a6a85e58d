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
#include "randperm_dev.h"
#include "match_cov_dev.h"
#include "pose_apply_dev.h"
#include "obs_filter_dev.h"
#include <math.h>

namespace {

__global__ __launch_bounds__(256) void frontend_epilogue_kernel(
    const float* __restrict__ flow, const float* __restrict__ logcov, int cov_is_log, int plane, float bl_fx,
    float bl_fx_sq, float* __restrict__ disparity, float* __restrict__ disparity_cov, float* __restrict__ depth,
    float* __restrict__ depth_cov, uint8_t* __restrict__ bad_mask, float* __restrict__ match_flow,
    float* __restrict__ match_cov) {
    MV_CHAIN_KERNEL_PRIO();
    // lane-batched (blockIdx.y = lane): inputs [lanes, 2, 2, H, W], every output [lanes, ch, H, W]
    const size_t lo = (size_t)blockIdx.y * plane;
    flow += 4 * lo; logcov += 4 * lo;
    if (disparity) disparity += lo;
    if (disparity_cov) disparity_cov += lo;
    if (depth) depth += lo;
    if (depth_cov) depth_cov += lo;
    if (bad_mask) bad_mask += lo;
    if (match_flow) match_flow += 2 * lo;
    if (match_cov) match_cov += 3 * lo;
    const mvEpiArgs ea{flow, logcov, cov_is_log, bl_fx, bl_fx_sq, disparity, disparity_cov, depth, depth_cov, match_flow, match_cov,
                       bad_mask};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += gridDim.x * blockDim.x) mv_epilogue_pixel(ea, plane, i);
}

// Lane-batched: blockIdx.y = lane; every per-keypoint table is [lanes, ..., cap] with `cap` rows of capacity per lane of
// which cnt.n[lane] are live (rows beyond are left untouched); every map is [lanes, ch, H, W].  lanes = 1, cap = N is the
// plain single-frame call.
// one keypoint of kp_track (pointers already offset to the lane; `vs` = row stride of the SoA value table)
__device__ __forceinline__ void kp_track_one(int n, int u0, int v0, const float* __restrict__ match_flow, const float* __restrict__ match_cov,
                                             const float* __restrict__ depth0, const float* __restrict__ disp0,
                                             const float* __restrict__ sdisp0, const float* __restrict__ sdd0,
                                             const float* __restrict__ depth1, const float* __restrict__ disp1,
                                             const float* __restrict__ sdisp1, const float* __restrict__ sdd1, int H, int W, int edge,
                                             float match_cov_default, float* __restrict__ out_kp0, float* __restrict__ out_kp1,
                                             uint8_t* __restrict__ out_inbound, float* __restrict__ out_vals, size_t vs,
                                             float* __restrict__ out_sigma0, float* __restrict__ out_sigma1) {
    const int plane = H * W;
    const bool ok0 = u0 >= 0 && u0 < W && v0 >= 0 && v0 < H;
    const int i0 = ok0 ? v0 * W + u0 : 0;
    // int64 + float32 -> float32 (torch type promotion)
    const float u1 = (float)u0 + match_flow[i0];
    const float v1 = (float)v0 + match_flow[plane + i0];
    const bool inb = ok0 && (u1 < (float)(W - edge)) && (u1 > (float)edge) && (v1 < (float)(H - edge)) && (v1 > (float)edge);
    if (out_kp0) { out_kp0[2 * n] = (float)u0; out_kp0[2 * n + 1] = (float)v0; }
    out_kp1[2 * n] = u1;
    out_kp1[2 * n + 1] = v1;
    out_inbound[n] = inb;
    float* o = out_vals + n;   // SoA: column k of lane l lives at out_vals[(k * lanes + l) * N + n]
    o[0 * vs] = depth0[i0];
    o[1 * vs] = disp0 ? disp0[i0] : -1.f;
    o[2 * vs] = sdisp0 ? sdisp0[i0] : -1.f;
    o[3 * vs] = sdd0 ? sdd0[i0] : -1.f;
    if (inb) {
        const int i1 = (int)v1 * W + (int)u1;  // .long(): truncation toward zero
        o[4 * vs] = depth1[i1];
        o[5 * vs] = disp1 ? disp1[i1] : -1.f;
        o[6 * vs] = sdisp1 ? sdisp1[i1] : -1.f;
        o[7 * vs] = sdd1 ? sdd1[i1] : -1.f;
    } else {
        o[4 * vs] = o[5 * vs] = o[6 * vs] = o[7 * vs] = 0.f;
    }
    // match covariance is read at the SOURCE pixel kp0 (MACVO.py:231)
    const float suu = match_cov ? match_cov[i0] : -1.f;
    const float svv = match_cov ? match_cov[plane + i0] : -1.f;
    const float suv = match_cov ? match_cov[2 * plane + i0] : -1.f;
    o[8 * vs] = suu; o[9 * vs] = svv; o[10 * vs] = suv;
    if (out_sigma1) { out_sigma1[3 * n] = suu; out_sigma1[3 * n + 1] = svv; out_sigma1[3 * n + 2] = suv; }
    // kp0 carries the constant quantisation sigma (MACVO.py:228-229)
    if (out_sigma0) { out_sigma0[3 * n] = match_cov_default; out_sigma0[3 * n + 1] = match_cov_default; out_sigma0[3 * n + 2] = 0.f; }
}

struct TrackArgs {   // per-lane base pointers of kp_track (offset by the kernels)
    const float *match_flow, *match_cov, *depth0, *disp0, *sdisp0, *sdd0, *depth1, *disp1, *sdisp1, *sdd1;
    int H, W, edge;
    float match_cov_default;
    float *out_kp0, *out_kp1;
    uint8_t* out_inbound;
    float *out_vals, *out_sigma0, *out_sigma1;
};
__device__ __forceinline__ void track_lane_offsets(TrackArgs& a, int lane, int N) {
    const size_t lp = (size_t)lane * a.H * a.W, ln = (size_t)lane * N;
    a.match_flow += 2 * lp;
    if (a.match_cov) a.match_cov += 3 * lp;
    a.depth0 += lp; a.depth1 += lp;
    if (a.disp0) a.disp0 += lp;
    if (a.sdisp0) a.sdisp0 += lp;
    if (a.sdd0) a.sdd0 += lp;
    if (a.disp1) a.disp1 += lp;
    if (a.sdisp1) a.sdisp1 += lp;
    if (a.sdd1) a.sdd1 += lp;
    if (a.out_kp0) a.out_kp0 += 2 * ln;
    a.out_kp1 += 2 * ln;
    a.out_inbound += ln;
    a.out_vals += ln;   // SoA table [11, lanes, cap]: row stride = lanes * cap
    if (a.out_sigma0) a.out_sigma0 += 3 * ln;
    if (a.out_sigma1) a.out_sigma1 += 3 * ln;
}

// Lane-batched: blockIdx.y = lane; every per-keypoint table is [lanes, ..., cap] with `cap` rows of capacity per lane of
// which cnt.n[lane] are live (rows beyond are left untouched); every map is [lanes, ch, H, W].  lanes = 1, cap = N is the
// plain single-frame call.
__global__ __launch_bounds__(256) void kp_track_kernel(const int64_t* __restrict__ kp0_uv, int N, mvLaneCounts cnt, TrackArgs a) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = blockIdx.y;
    if (n >= cnt.n[lane]) return;
    kp0_uv += 2 * (size_t)lane * N;
    track_lane_offsets(a, lane, N);
    kp_track_one(n, (int)kp0_uv[2 * n], (int)kp0_uv[2 * n + 1], a.match_flow, a.match_cov, a.depth0, a.disp0, a.sdisp0, a.sdd0, a.depth1,
                 a.disp1, a.sdisp1, a.sdd1, a.H, a.W, a.edge, a.match_cov_default, a.out_kp0, a.out_kp1, a.out_inbound, a.out_vals,
                 (size_t)gridDim.y * N, a.out_sigma0, a.out_sigma1);
}

// Round 3: gather + track + camera-frame back-projection of a frame's keypoints in ONE launch (mv_kp_front_lanes).  On the
// backend stream of a one-lane pipeline every launch boundary costs ~5 us beside the volume GEMM, and the three kernels are
// per-keypoint maps over the same 200 rows: `selected[perm][..., 2:].roll(1, 1)` (KeypointSelector.py:331-332,404-405), the
// tracking gathers of MACVO.py:198-232 and pixel2point_NED (Utility/Point.py:15-17) — same expressions, same bits.  With one
// lane the permutation travels INSIDE the kernel arguments (<= 256 indices = 1 KB of the 4 KB kernarg segment): no pinned
// staging copy, no host-to-device memcpy node on the stream.
struct PermArg {
    int32_t idx[256];
};
template <bool PERM_IN_ARGS>
__global__ __launch_bounds__(256) void kp_front_kernel(const int32_t* __restrict__ cand, size_t cand_lane_stride,
                                                       const int64_t* __restrict__ perm, PermArg pa, int cap, mvLaneCounts cnt,
                                                       int64_t* __restrict__ out_uv, TrackArgs a, float fx, float fy, float cx, float cy,
                                                       float* __restrict__ pos_Tc) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = blockIdx.y;
    if (n >= cnt.n[lane]) return;
    cand += (size_t)lane * cand_lane_stride;
    out_uv += 2 * (size_t)lane * cap;
    pos_Tc += 3 * (size_t)lane * cap;
    track_lane_offsets(a, lane, cap);
    const long pi = PERM_IN_ARGS ? (long)pa.idx[n] : (long)perm[(size_t)lane * cap + n];
    const int lin = cand[pi];
    const int u0 = lin % a.W, v0 = lin / a.W;
    out_uv[2 * n + 0] = u0;
    out_uv[2 * n + 1] = v0;
    const size_t vs = (size_t)gridDim.y * cap;
    kp_track_one(n, u0, v0, a.match_flow, a.match_cov, a.depth0, a.disp0, a.sdisp0, a.sdd0, a.depth1, a.disp1, a.sdisp1, a.sdd1, a.H,
                 a.W, a.edge, a.match_cov_default, a.out_kp0, a.out_kp1, a.out_inbound, a.out_vals, vs, a.out_sigma0, a.out_sigma1);
    // pixel2point_NED on (kp0, depth0 at kp0): pp.pixel2point -> (((u-cx)*d)/fx, ((v-cy)*d)/fy, d) rolled to (d, x, y)
    const float u = (float)u0, v = (float)v0, d = a.out_vals[n];
    pos_Tc[3 * n] = d;
    pos_Tc[3 * n + 1] = ((u - cx) * d) / fx;
    pos_Tc[3 * n + 2] = ((v - cy) * d) / fy;
}

__global__ __launch_bounds__(256) void backproject_kernel(const float* __restrict__ kp_uv,
                                                          const float* __restrict__ depth_vals, int depth_stride,
                                                          float fx, float fy, float cx, float cy,
                                                          const float* __restrict__ pose, int cap, mvLaneCounts cnt,
                                                          size_t depth_lane_stride,
                                                          float* __restrict__ pos_Tc, float* __restrict__ pos_Tw,
                                                          double* __restrict__ rot) {
    // lane-batched (blockIdx.y = lane): kp_uv / pos_* are [lanes, cap, .], pose [lanes, 7], rot [lanes, 9]
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = blockIdx.y;
    const int N = cnt.n[lane];
    {
        const size_t ln = (size_t)lane * cap;
        if (kp_uv) kp_uv += 2 * ln;
        if (depth_vals) depth_vals += (size_t)lane * depth_lane_stride;
        if (pose) pose += 7 * lane;
        if (pos_Tc) pos_Tc += 3 * ln;
        if (pos_Tw) pos_Tw += 3 * ln;
        if (rot) rot += 9 * lane;
    }
    float t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1};
    if (pose) {
        t[0] = pose[0]; t[1] = pose[1]; t[2] = pose[2];
        q[0] = pose[3]; q[1] = pose[4]; q[2] = pose[5]; q[3] = pose[6];
    }
    if (n == 0 && rot && pose) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float e[3] = {c == 0 ? 1.f : 0.f, c == 1 ? 1.f : 0.f, c == 2 ? 1.f : 0.f};
            float col[3];
            quat_act_f32(q, e, col);
            rot[0 * 3 + c] = (double)col[0];
            rot[1 * 3 + c] = (double)col[1];
            rot[2 * 3 + c] = (double)col[2];
        }
    }
    if (n >= N) return;
    // pixel2point_NED: pp.pixel2point -> (((u-cx)*d)/fx, ((v-cy)*d)/fy, d) rolled to (d, x, y)
    const float u = kp_uv[2 * n], v = kp_uv[2 * n + 1], d = depth_vals[(size_t)n * depth_stride];
    const float p[3] = {d, ((u - cx) * d) / fx, ((v - cy) * d) / fy};
    if (pos_Tc) { pos_Tc[3 * n] = p[0]; pos_Tc[3 * n + 1] = p[1]; pos_Tc[3 * n + 2] = p[2]; }
    if (pos_Tw) {
        float r[3];
        quat_act_f32(q, p, r);  // SE3 Act = SO3 Act + t
        pos_Tw[3 * n] = r[0] + t[0]; pos_Tw[3 * n + 1] = r[1] + t[1]; pos_Tw[3 * n + 2] = r[2] + t[2];
    }
}

// The pose-dependent remainder of the backend, split off so that everything else of a frame's backend can run BEFORE the
// previous frame's solve has finished (Odometry/MACVO.py:273-281): pos_Tw = T_prev * p_cam (fp32, PyPose SE3 Act),
// rot = R_prev (fp32 -> fp64, as backproject_kernel writes it) and cov_Tw = R cov_Tc R^T (fp64, the grouping of match_cov_kernel).
// Bit-identical to mv_backproject(..., pose, ...) + mv_match_cov(..., rot, ..., out_cov_rot) on the same inputs.
__global__ __launch_bounds__(256) void pose_apply_kernel(const float* __restrict__ pose, const float* __restrict__ pos_Tc,
                                                         const double* __restrict__ cov, int cap, mvLaneCounts cnt,
                                                         float* __restrict__ pos_Tw, double* __restrict__ rot,
                                                         double* __restrict__ cov_rot) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = blockIdx.y;
    const int N = cnt.n[lane];
    {
        const size_t ln = (size_t)lane * cap;
        pose += 7 * lane;
        if (pos_Tc) pos_Tc += 3 * ln;
        if (pos_Tw) pos_Tw += 3 * ln;
        if (cov) cov += 9 * ln;
        if (cov_rot) cov_rot += 9 * ln;
        if (rot) rot += 9 * lane;
    }
    double R[9];
    mv_pose_rotation(pose, R);
    if (n == 0 && rot) {
#pragma unroll
        for (int i = 0; i < 9; ++i) rot[i] = R[i];
    }
    if (n >= N) return;
    mv_pose_apply_row(pose, R, n, pos_Tc, cov, pos_Tw, cov_rot);
}

// Dense-mapping tail of run_pair (Odometry/MACVO.py:315-325,334): per selected map pixel gather depth and depth variance,
// back-project (pixel2point_NED), move to the world with the previous pose, and emit the constant match sigma.
__global__ __launch_bounds__(256) void map_points_kernel(const int64_t* __restrict__ uv, int N,
                                                          const float* __restrict__ depth, const float* __restrict__ depth_cov,
                                                          const float* __restrict__ image, int H, int W, float fx, float fy,
                                                          float cx, float cy, const float* __restrict__ pose,
                                                          float match_cov_default, float* __restrict__ out_uv,
                                                          float* __restrict__ out_d, float* __restrict__ out_sdd,
                                                          float* __restrict__ out_sigma, float* __restrict__ out_Tc,
                                                          float* __restrict__ out_Tw, uint8_t* __restrict__ out_color) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int u = (int)uv[2 * n], v = (int)uv[2 * n + 1];
    const bool ok = u >= 0 && u < W && v >= 0 && v < H;
    const int i = ok ? v * W + u : 0;
    const float d = depth[i];
    const float uf = (float)u, vf = (float)v;
    out_uv[2 * n] = uf;
    out_uv[2 * n + 1] = vf;
    if (out_d) out_d[n] = d;
    if (out_sdd) out_sdd[n] = depth_cov ? depth_cov[i] : -1.f;
    if (out_sigma) { out_sigma[3 * n] = match_cov_default; out_sigma[3 * n + 1] = match_cov_default; out_sigma[3 * n + 2] = 0.f; }
    const float pc[3] = {d, ((uf - cx) * d) / fx, ((vf - cy) * d) / fy};
    if (out_Tc) { out_Tc[3 * n] = pc[0]; out_Tc[3 * n + 1] = pc[1]; out_Tc[3 * n + 2] = pc[2]; }
    if (out_Tw) {
        const float q[4] = {pose[3], pose[4], pose[5], pose[6]};
        float r[3];
        quat_act_f32(q, pc, r);
        out_Tw[3 * n] = r[0] + pose[0]; out_Tw[3 * n + 1] = r[1] + pose[1]; out_Tw[3 * n + 2] = r[2] + pose[2];
    }
    if (out_color && image) {   // (imageL * 255).to(uint8): truncation toward zero of a [0,1] float (MACVO.py:327-328)
        const size_t plane = (size_t)H * W;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = image[c * plane + i] * 255.f;
            out_color[3 * n + c] = (uint8_t)(int)x;
        }
    }
}

__global__ __launch_bounds__(256) void obs_filter_kernel(const uint8_t* __restrict__ inbound,
                                                         const double* __restrict__ cov1,
                                                         const double* __restrict__ cov2,
                                                         const float* __restrict__ vals, int flags,
                                                         float min_depth, float max_depth, int cap, mvLaneCounts cnt,
                                                         uint8_t* __restrict__ valid, int32_t* __restrict__ count) {
    obs_filter_body(inbound, cov1, cov2, vals, flags, min_depth, max_depth, cap, cnt.n[blockIdx.x], blockIdx.x, gridDim.x, valid, count);
}

// VERDICT r4 next #3 — the pose-INDEPENDENT half of a frame's backend in ONE launch: gather + track + back-projection (kp_front_kernel) and
// both 31 x 31 covariance models (match_cov_kernel).  Odometry/MACVO.py:197-262.  On the backend stream of a one-lane pipeline these were launches of
// <= 200-point work whose boundaries (each a dependent dispatch beside a GEMM that owns every CU) cost more than their arithmetic.
//   grid (ceil(n_max / 4), 2 keypoint sets, lanes), 4 waves: a wave = one keypoint of one set.  Every wave re-derives the keypoint's tracked
//   position and match sigma itself (five uniform loads: cheaper than a hand-over through memory); the set-0 wave's lane 0 writes the tracking
//   tables, each wave its own sigma row and covariance.
// The observation filters need every covariance of the lane.  A first version ran them in the last workgroup to finish (ticket counter, __threadfence):
// bit-identical, but SLOWER in the pipe (5.17-5.21 k vs 5.41-5.56 k frames/s, profiles/r05_pipe_ab.log) — an agent-scope release / acquire on this
// 8-XCD part writes back and invalidates the L2s, 200 times per frame, under the GEMM that runs beside it (its launches went from 87-94 to 98-103 us).
// The filters are now the PROLOGUE of the lane's solve workgroup (mv_pgo_solve_posed), behind the kernel boundary that already exists.
// Same expressions, same order, same bits as the separate kernels.
// Round 6 — PM = 2, the device-driven frame: the keypoint permutation `torch.randperm(count)[:num_point]` (KeypointSelector.py:331,404) is drawn HERE, from the
// candidate count where the selector left it in device memory and a device-resident MT19937 (randperm_dev.h): no D2H count, no host generator, no H2D
// permutation, no host wait anywhere in a frame.  Every workgroup draws the head for itself (LDS); workgroup (0, 0, lane) also advances the lane's
// generator into the other state buffer and publishes the head and the live-row count for the solve (same stream, next launch) and for result views.
struct DrawArgs {
    const int32_t* count;      // [lanes, count_stride]: candidate count of lane l at count[l * count_stride]
    int count_stride;
    const uint32_t* state_in;  // [lanes, mvrp::MT_STRIDE]
    uint32_t* state_out;       // [lanes, mvrp::MT_STRIDE], != state_in
    int k;                     // num_point
    int64_t* out_perm;         // [lanes, cap]
    int32_t* out_live;         // [lanes, 2]: selected keypoints = min(count, k), candidate count
};
struct DrawLds {
    mvrp::Scratch s;
    int32_t head[mvrp::NBUCKET];
    int32_t out[mvrp::MAX_HEAD];
};
template <int PM>   // where the permutation comes from: 0 device memory, 1 the kernel arguments, 2 drawn here
__global__ __launch_bounds__(256) void backend_front_kernel(const int32_t* __restrict__ cand, size_t cand_lane_stride, const int64_t* __restrict__ perm,
                                                            PermArg pa, int cap, mvLaneCounts cnt, int64_t* out_uv, TrackArgs a, float fx, float fy,
                                                            float cx, float cy, float* pos_Tc, const float* depth_map0, const float* depth_map1,
                                                            double* out_cov0, double* out_cov1, mvMatchCovParams cp, DrawArgs da) {
    __shared__ __attribute__((aligned(16))) char draw_raw[PM == 2 ? sizeof(DrawLds) : 16];
    DrawLds& D = *reinterpret_cast<DrawLds*>(draw_raw);
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int set = blockIdx.y, pl = blockIdx.z;
    int N;
    if constexpr (PM == 2) {
        const int64_t n_cand = da.count[(size_t)pl * da.count_stride];
        const bool adv = blockIdx.x == 0 && blockIdx.y == 0;
        N = mv_randperm_head_wg(da.state_in + (size_t)pl * mvrp::MT_STRIDE, adv ? da.state_out + (size_t)pl * mvrp::MT_STRIDE : nullptr, n_cand, da.k,
                                D.out, D.s, D.head);
        __syncthreads();
        if (adv) {
            for (int i = threadIdx.x; i < N; i += 256) da.out_perm[(size_t)pl * cap + i] = D.out[i];
            if (threadIdx.x == 0) { da.out_live[2 * pl] = N; da.out_live[2 * pl + 1] = (int32_t)n_cand; }
        }
    } else {
        N = cnt.n[pl];
    }
    if (n < N) {       // (wave-uniform)
        const int32_t* cand_l = cand + (size_t)pl * cand_lane_stride;
        const long pi = PM == 2 ? (long)D.out[n] : PM == 1 ? (long)pa.idx[n] : (long)perm[(size_t)pl * cap + n];
        const int lin = cand_l[pi];
        const int u0 = lin % a.W, v0 = lin / a.W;
        TrackArgs al = a;
        track_lane_offsets(al, pl, cap);
        const size_t ln = (size_t)pl * cap;
        float u, v, suu, svv, suv;
        if (set == 0) {
            if (lane == 0) {     // kp_front_kernel's body; sigma1 is the set-1 wave's row
                out_uv[2 * (ln + n) + 0] = u0;
                out_uv[2 * (ln + n) + 1] = v0;
                kp_track_one(n, u0, v0, al.match_flow, al.match_cov, al.depth0, al.disp0, al.sdisp0, al.sdd0, al.depth1, al.disp1, al.sdisp1, al.sdd1,
                             al.H, al.W, al.edge, al.match_cov_default, al.out_kp0, al.out_kp1, al.out_inbound, al.out_vals, (size_t)gridDim.z * cap,
                             nullptr, nullptr);
                al.out_sigma0[3 * n + 2] = 0.f;                                   // ([3n], [3n + 1]: written (clamped) by the covariance model below)
                const float uf = (float)u0, vf = (float)v0, d = al.out_vals[n];
                float* pt = pos_Tc + 3 * (ln + n);
                pt[0] = d;
                pt[1] = ((uf - cx) * d) / fx;
                pt[2] = ((vf - cy) * d) / fy;
            }
            u = (float)u0; v = (float)v0;                                         // kp0 carries the constant quantisation sigma (MACVO.py:228-229)
            suu = a.match_cov_default; svv = a.match_cov_default; suv = 0.f;
        } else {
            const int plane = al.H * al.W;
            const bool ok0 = u0 >= 0 && u0 < al.W && v0 >= 0 && v0 < al.H;
            const int i0 = ok0 ? v0 * al.W + u0 : 0;
            u = (float)u0 + al.match_flow[i0];                                    // kp_track_one's u1, v1 and match sigma, re-derived
            v = (float)v0 + al.match_flow[plane + i0];
            suu = al.match_cov ? al.match_cov[i0] : -1.f;
            svv = al.match_cov ? al.match_cov[plane + i0] : -1.f;
            suv = al.match_cov ? al.match_cov[2 * plane + i0] : -1.f;
            if (lane == 0) al.out_sigma1[3 * n + 2] = suv;                        // ([3n], [3n + 1]: written clamped by the covariance model below)
        }
        const mvcov::CovSet S{set ? depth_map1 : depth_map0, nullptr, set ? a.out_sigma1 : a.out_sigma0, nullptr, nullptr, set ? out_cov1 : out_cov0,
                              nullptr, nullptr};
        mvcov::match_cov_wave_vals(S, cp, cap, pl, n, u, v, suu, svv, suv);
    }
}

}  // namespace

static inline int check_lanes(int lanes, const int32_t* n_live, int cap, mvLaneCounts& c, int& n_max) {
    MV_CHECK_ARG(lanes >= 1 && lanes <= MV_MAX_LANES && n_live && cap >= 0);
    n_max = 0;
    for (int l = 0; l < lanes; ++l) {
        MV_CHECK_ARG(n_live[l] >= 0 && n_live[l] <= cap);
        c.n[l] = n_live[l];
        n_max = n_live[l] > n_max ? n_live[l] : n_max;
    }
    return MV_OK;
}

extern "C" int mv_frontend_epilogue_lanes(const float* flow, const float* logcov, int cov_is_log, int H, int W,
                                          float bl_fx, float bl_fx_sq, float* disparity, float* disparity_cov,
                                          float* depth, float* depth_cov, uint8_t* bad_mask, float* match_flow,
                                          float* match_cov, int lanes, mvStream_t stream) {
    MV_CHECK_ARG(flow && logcov && H > 0 && W > 0 && lanes >= 1 && lanes <= MV_MAX_LANES);
    const int plane = H * W;
    const int blocks = min(mv_ceil_div(plane, 256), 2048);
    hipLaunchKernelGGL(frontend_epilogue_kernel, dim3(blocks, lanes), dim3(256), 0, (hipStream_t)stream, flow, logcov,
                       cov_is_log, plane, bl_fx, bl_fx_sq, disparity, disparity_cov, depth, depth_cov, bad_mask,
                       match_flow, match_cov);
    return mv_launch_status();
}

extern "C" int mv_frontend_epilogue(const float* flow, const float* logcov, int cov_is_log, int H, int W,
                                    float bl_fx, float bl_fx_sq, float* disparity, float* disparity_cov, float* depth,
                                    float* depth_cov, uint8_t* bad_mask, float* match_flow, float* match_cov,
                                    mvStream_t stream) {
    return mv_frontend_epilogue_lanes(flow, logcov, cov_is_log, H, W, bl_fx, bl_fx_sq, disparity, disparity_cov, depth,
                                      depth_cov, bad_mask, match_flow, match_cov, 1, stream);
}

extern "C" int mv_kp_track_lanes(const int64_t* kp0_uv, int lanes, const int32_t* n_live, int cap, const float* match_flow,
                                 const float* match_cov, const float* depth0, const float* disp0, const float* sdisp0,
                                 const float* sdd0, const float* depth1, const float* disp1, const float* sdisp1,
                                 const float* sdd1, int H, int W, int edge, float match_cov_default, float* out_kp0,
                                 float* out_kp1, uint8_t* out_inbound, float* out_vals, float* out_sigma0,
                                 float* out_sigma1, mvStream_t stream) {
    MV_CHECK_ARG(H > 0 && W > 0 && edge >= 0);
    mvLaneCounts c{};
    int n_max = 0;
    const int rc = check_lanes(lanes, n_live, cap, c, n_max);
    if (rc != MV_OK) return rc;
    if (n_max == 0) return MV_OK;
    MV_CHECK_ARG(kp0_uv && match_flow && depth0 && depth1 && out_kp1 && out_inbound && out_vals);
    const TrackArgs ta{match_flow, match_cov, depth0, disp0, sdisp0, sdd0, depth1, disp1, sdisp1, sdd1, H, W, edge, match_cov_default,
                       out_kp0, out_kp1, out_inbound, out_vals, out_sigma0, out_sigma1};
    hipLaunchKernelGGL(kp_track_kernel, dim3(mv_ceil_div(n_max, 256), lanes), dim3(256), 0, (hipStream_t)stream, kp0_uv, cap, c, ta);
    return mv_launch_status();
}

extern "C" int mv_kp_front_lanes(const int32_t* cand, size_t cand_lane_stride, const int64_t* perm_dev, const int64_t* perm_host,
                                 int lanes, const int32_t* n_live, int cap, const float* match_flow, const float* match_cov,
                                 const float* depth0, const float* disp0, const float* sdisp0, const float* sdd0,
                                 const float* depth1, const float* disp1, const float* sdisp1, const float* sdd1, int H, int W,
                                 int edge, float match_cov_default, float fx, float fy, float cx, float cy, int64_t* out_kp0_uv,
                                 float* out_kp0, float* out_kp1, uint8_t* out_inbound, float* out_vals, float* out_sigma0,
                                 float* out_sigma1, float* out_pos_Tc, mvStream_t stream) {
    MV_CHECK_ARG(H > 0 && W > 0 && edge >= 0);
    mvLaneCounts c{};
    int n_max = 0;
    const int rc = check_lanes(lanes, n_live, cap, c, n_max);
    if (rc != MV_OK) return rc;
    if (n_max == 0) return MV_OK;
    MV_CHECK_ARG(cand && (perm_dev || perm_host) && match_flow && depth0 && depth1 && out_kp0_uv && out_kp1 && out_inbound && out_vals &&
                 out_pos_Tc);
    const TrackArgs ta{match_flow, match_cov, depth0, disp0, sdisp0, sdd0, depth1, disp1, sdisp1, sdd1, H, W, edge, match_cov_default,
                       out_kp0, out_kp1, out_inbound, out_vals, out_sigma0, out_sigma1};
    const dim3 grid(mv_ceil_div(n_max, 256), lanes), block(256);
    if (perm_host && lanes == 1 && n_max <= 256) {     // the permutation rides in the kernel arguments
        PermArg pa;
        for (int i = 0; i < n_max; ++i) {
            MV_CHECK_ARG(perm_host[i] >= 0 && perm_host[i] <= 0x7fffffffLL);
            pa.idx[i] = (int32_t)perm_host[i];
        }
        hipLaunchKernelGGL(kp_front_kernel<true>, grid, block, 0, (hipStream_t)stream, cand, cand_lane_stride, nullptr, pa, cap, c,
                           out_kp0_uv, ta, fx, fy, cx, cy, out_pos_Tc);
    } else {
        MV_CHECK_ARG(perm_dev);
        PermArg pa;
        pa.idx[0] = 0;
        hipLaunchKernelGGL(kp_front_kernel<false>, grid, block, 0, (hipStream_t)stream, cand, cand_lane_stride, perm_dev, pa, cap, c,
                           out_kp0_uv, ta, fx, fy, cx, cy, out_pos_Tc);
    }
    return mv_launch_status();
}

extern "C" int mv_backend_front_lanes(const int32_t* cand, size_t cand_lane_stride, const int64_t* perm_dev, const int64_t* perm_host, int lanes,
                                      const int32_t* n_live, int cap, const float* match_flow, const float* match_cov, const float* depth0,
                                      const float* disp0, const float* sdisp0, const float* sdd0, const float* depth1, const float* disp1,
                                      const float* sdisp1, const float* sdd1, int edge, float match_cov_default, const mvMatchCovParams* cov_params,
                                      int64_t* out_kp0_uv, float* out_kp0, float* out_kp1, uint8_t* out_inbound, float* out_vals, float* out_sigma0,
                                      float* out_sigma1, float* out_pos_Tc, double* out_cov0, double* out_cov1, mvStream_t stream) {
    MV_CHECK_ARG(cov_params && edge >= 0);
    const mvMatchCovParams cp = *cov_params;
    MV_CHECK_ARG(cp.H > 0 && cp.W > 0 && cp.kernel_size >= 1 && (cp.kernel_size & 1) && cp.use_patch_var);
    if (cp.kernel_size > mvcov::MAX_K) return MV_ERR_UNSUPPORTED;
    mvLaneCounts c{};
    int n_max = 0;
    const int rc = check_lanes(lanes, n_live, cap, c, n_max);
    if (rc != MV_OK) return rc;
    if (n_max == 0) return MV_OK;
    MV_CHECK_ARG(cand && (perm_dev || perm_host) && match_flow && depth0 && depth1 && out_kp0_uv && out_kp1 && out_inbound && out_vals &&
                 out_sigma0 && out_sigma1 && out_pos_Tc && out_cov0 && out_cov1);
    const TrackArgs ta{match_flow, match_cov, depth0, disp0, sdisp0, sdd0, depth1, disp1, sdisp1, sdd1, cp.H, cp.W, edge, match_cov_default,
                       out_kp0, out_kp1, out_inbound, out_vals, out_sigma0, out_sigma1};
    const dim3 grid(mv_ceil_div(n_max, 4), 2, lanes), block(256);
    if (perm_host && lanes == 1 && n_max <= 256) {     // the permutation rides in the kernel arguments
        PermArg pa;
        for (int i = 0; i < n_max; ++i) {
            MV_CHECK_ARG(perm_host[i] >= 0 && perm_host[i] <= 0x7fffffffLL);
            pa.idx[i] = (int32_t)perm_host[i];
        }
        hipLaunchKernelGGL(backend_front_kernel<1>, grid, block, 0, (hipStream_t)stream, cand, cand_lane_stride, nullptr, pa, cap, c, out_kp0_uv, ta,
                           cp.fx, cp.fy, cp.cx, cp.cy, out_pos_Tc, depth0, depth1, out_cov0, out_cov1, cp, DrawArgs{});
    } else {
        MV_CHECK_ARG(perm_dev);
        PermArg pa;
        pa.idx[0] = 0;
        hipLaunchKernelGGL(backend_front_kernel<0>, grid, block, 0, (hipStream_t)stream, cand, cand_lane_stride, perm_dev, pa, cap, c, out_kp0_uv, ta,
                           cp.fx, cp.fy, cp.cx, cp.cy, out_pos_Tc, depth0, depth1, out_cov0, out_cov1, cp, DrawArgs{});
    }
    return mv_launch_status();
}

// mv_backend_front_lanes of the device-driven frame (round 6): the permutation is drawn inside the launch (backend_front_kernel<2>), the number of live rows
// never reaches the host — the grid covers `num_point` rows per lane and the waves beyond min(count, num_point) retire at once.
extern "C" int mv_backend_front_draw_lanes(const int32_t* cand, size_t cand_lane_stride, const int32_t* count_dev, int count_stride, const uint32_t* state_in,
                                           uint32_t* state_out, int num_point, int lanes, int cap, const float* match_flow, const float* match_cov,
                                           const float* depth0, const float* disp0, const float* sdisp0, const float* sdd0, const float* depth1,
                                           const float* disp1, const float* sdisp1, const float* sdd1, int edge, float match_cov_default,
                                           const mvMatchCovParams* cov_params, int64_t* out_perm, int32_t* out_live, int64_t* out_kp0_uv, float* out_kp0,
                                           float* out_kp1, uint8_t* out_inbound, float* out_vals, float* out_sigma0, float* out_sigma1, float* out_pos_Tc,
                                           double* out_cov0, double* out_cov1, mvStream_t stream) {
    MV_CHECK_ARG(cov_params && edge >= 0 && lanes >= 1 && lanes <= MV_MAX_LANES && cap >= 1 && num_point >= 0 && num_point <= cap && count_stride >= 1);
    const mvMatchCovParams cp = *cov_params;
    MV_CHECK_ARG(cp.H > 0 && cp.W > 0 && cp.kernel_size >= 1 && (cp.kernel_size & 1) && cp.use_patch_var);
    if (cp.kernel_size > mvcov::MAX_K || num_point > mvrp::MAX_HEAD) return MV_ERR_UNSUPPORTED;
    MV_CHECK_ARG(cand && count_dev && state_in && state_out && state_in != state_out && out_perm && out_live && match_flow && depth0 && depth1 &&
                 out_kp0_uv && out_kp1 && out_inbound && out_vals && out_sigma0 && out_sigma1 && out_pos_Tc && out_cov0 && out_cov1);
    const TrackArgs ta{match_flow, match_cov, depth0, disp0, sdisp0, sdd0, depth1, disp1, sdisp1, sdd1, cp.H, cp.W, edge, match_cov_default,
                       out_kp0, out_kp1, out_inbound, out_vals, out_sigma0, out_sigma1};
    const dim3 grid(mv_ceil_div(num_point > 0 ? num_point : 1, 4), 2, lanes), block(256);   // (num_point == 0: workgroup (0, 0, l) still advances the generator)
    PermArg pa;
    pa.idx[0] = 0;
    const DrawArgs da{count_dev, count_stride, state_in, state_out, num_point, out_perm, out_live};
    hipLaunchKernelGGL(backend_front_kernel<2>, grid, block, 0, (hipStream_t)stream, cand, cand_lane_stride, nullptr, pa, cap, mvLaneCounts{}, out_kp0_uv, ta,
                       cp.fx, cp.fy, cp.cx, cp.cy, out_pos_Tc, depth0, depth1, out_cov0, out_cov1, cp, da);
    return mv_launch_status();
}

extern "C" int mv_kp_track(const int64_t* kp0_uv, int N, const float* match_flow, const float* match_cov,
                           const float* depth0, const float* disp0, const float* sdisp0, const float* sdd0,
                           const float* depth1, const float* disp1, const float* sdisp1, const float* sdd1, int H,
                           int W, int edge, float match_cov_default, float* out_kp0, float* out_kp1,
                           uint8_t* out_inbound, float* out_vals, float* out_sigma0, float* out_sigma1,
                           mvStream_t stream) {
    MV_CHECK_ARG(N >= 0);
    const int32_t n = N;
    return mv_kp_track_lanes(kp0_uv, 1, &n, N, match_flow, match_cov, depth0, disp0, sdisp0, sdd0, depth1, disp1, sdisp1,
                             sdd1, H, W, edge, match_cov_default, out_kp0, out_kp1, out_inbound, out_vals, out_sigma0,
                             out_sigma1, stream);
}

extern "C" int mv_backproject_lanes(const float* kp_uv, const float* depth_vals, int depth_stride, size_t depth_lane_stride,
                                    float fx, float fy, float cx, float cy, const float* pose, int lanes,
                                    const int32_t* n_live, int cap, float* pos_Tc, float* pos_Tw, double* rot,
                                    mvStream_t stream) {
    MV_CHECK_ARG(depth_stride >= 1);
    MV_CHECK_ARG((!pos_Tw && !rot) || pose);
    mvLaneCounts c{};
    int n_max = 0;
    const int rc = check_lanes(lanes, n_live, cap, c, n_max);
    if (rc != MV_OK) return rc;
    if (n_max == 0 && !rot) return MV_OK;
    MV_CHECK_ARG(n_max == 0 || (kp_uv && depth_vals));
    hipLaunchKernelGGL(backproject_kernel, dim3(n_max > 0 ? mv_ceil_div(n_max, 256) : 1, lanes), dim3(256), 0,
                       (hipStream_t)stream, kp_uv, depth_vals, depth_stride, fx, fy, cx, cy, pose, cap, c,
                       depth_lane_stride, pos_Tc, pos_Tw, rot);
    return mv_launch_status();
}

extern "C" int mv_pose_apply_lanes(const float* pose, const float* pos_Tc, const double* cov, int lanes, const int32_t* n_live,
                                   int cap, float* pos_Tw, double* rot, double* cov_rot, mvStream_t stream) {
    MV_CHECK_ARG(pose);
    MV_CHECK_ARG((pos_Tw == nullptr) == (pos_Tc == nullptr) && (cov_rot == nullptr) == (cov == nullptr));
    mvLaneCounts c{};
    int n_max = 0;
    const int rc = check_lanes(lanes, n_live, cap, c, n_max);
    if (rc != MV_OK) return rc;
    if (n_max == 0 && !rot) return MV_OK;
    hipLaunchKernelGGL(pose_apply_kernel, dim3(n_max > 0 ? mv_ceil_div(n_max, 256) : 1, lanes), dim3(256), 0, (hipStream_t)stream,
                       pose, pos_Tc, cov, cap, c, pos_Tw, rot, cov_rot);
    return mv_launch_status();
}

extern "C" int mv_backproject(const float* kp_uv, const float* depth_vals, int depth_stride, float fx, float fy,
                              float cx, float cy, const float* pose, int N, float* pos_Tc, float* pos_Tw, double* rot,
                              mvStream_t stream) {
    MV_CHECK_ARG(N >= 0);
    const int32_t n = N;
    return mv_backproject_lanes(kp_uv, depth_vals, depth_stride, 0, fx, fy, cx, cy, pose, 1, &n, N, pos_Tc, pos_Tw, rot,
                                stream);
}

extern "C" int mv_obs_filter_lanes(const uint8_t* inbound, const double* cov1, const double* cov2, const float* vals,
                                   int flags, float min_depth, float max_depth, int lanes, const int32_t* n_live, int cap,
                                   uint8_t* valid, int32_t* count, mvStream_t stream) {
    MV_CHECK_ARG(count);
    mvLaneCounts c{};
    int n_max = 0;
    const int rc = check_lanes(lanes, n_live, cap, c, n_max);
    if (rc != MV_OK) return rc;
    if (cap > 0) {
        MV_CHECK_ARG(valid);
        MV_CHECK_ARG(n_max == 0 || !(flags & 1) || (cov1 && cov2));
        MV_CHECK_ARG(n_max == 0 || !(flags & 6) || vals);
    }
    hipLaunchKernelGGL(obs_filter_kernel, dim3(lanes), dim3(256), 0, (hipStream_t)stream, inbound, cov1, cov2, vals,
                       flags, min_depth, max_depth, cap, c, valid, count);
    return mv_launch_status();
}

extern "C" int mv_obs_filter(const uint8_t* inbound, const double* cov1, const double* cov2, const float* vals,
                             int flags, float min_depth, float max_depth, int N, uint8_t* valid, int32_t* count,
                             mvStream_t stream) {
    MV_CHECK_ARG(N >= 0);
    const int32_t n = N;
    return mv_obs_filter_lanes(inbound, cov1, cov2, vals, flags, min_depth, max_depth, 1, &n, N, valid, count, stream);
}

extern "C" int mv_map_points(const int64_t* uv, int N, const float* depth, const float* depth_cov, const float* image, int H,
                             int W, float fx, float fy, float cx, float cy, const float* pose, float match_cov_default,
                             float* out_uv, float* out_d, float* out_sdd, float* out_sigma, float* out_Tc, float* out_Tw,
                             uint8_t* out_color, mvStream_t stream) {
    MV_CHECK_ARG(N >= 0 && H > 0 && W > 0);
    if (N == 0) return MV_OK;
    MV_CHECK_ARG(uv && depth && out_uv);
    MV_CHECK_ARG(!out_Tw || pose);
    hipLaunchKernelGGL(map_points_kernel, dim3(mv_ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, uv, N, depth, depth_cov,
                       image, H, W, fx, fy, cx, cy, pose, match_cov_default, out_uv, out_d, out_sdd, out_sigma, out_Tc, out_Tw,
                       out_color);
    return mv_launch_status();
}
