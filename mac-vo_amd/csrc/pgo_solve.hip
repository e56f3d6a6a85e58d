// A17-A22 — covariance-weighted two-frame pose-graph solve on SE(3), one workgroup per problem
// (SURVEY.md §8 A17-A22).
//
// Replaces, for the newest-frame pose (the only variable, Graphs.py:83):
//   TwoFrame_PGO._optimize            Module/Optimization/TwoFramePGO/Optimizer.py:81-102
//   residual graphs + analytic J      Module/Optimization/TwoFramePGO/Graphs.py:33-231
//   LM_analytic.step                  Module/Optimization/PyposeOptimizers.py:160-194
//   PyPose 0.6.8 Huber / FastTriggs / RobustModel.loss / TrustRegion / PINV / StopOnPlateau / SE3 add_
// The per-point / per-pose arithmetic lives in pgo_math.h (shared with the host twin the CPU suite replays); this file is the
// kernel: who owns which point, the reductions, the LM loop and its reject rounds.
//
// What the reference materialises and this kernel does not:
//   * the dense 3N x 3N block_diag weight (2.9 MB fp64 for N = 200, rebuilt every outer iteration):
//     here each point applies its own 3x3 / 2x2 information block in registers;
//   * J [3N,7] and the 7x600 @ 600x600 matmul: here J_i^T W_i J_i (21 unique entries) and J_i^T W_i r_i (6) are formed per thread,
//     tree-reduced inside each wavefront with DPP butterflies and combined across the workgroup's 4 waves through LDS;
//   * the dead 7th tangent column (clamped to 1e-6, b_7 = 0 => D_7 = 0) is dropped analytically;
//     the 6x6 SPD system is solved by an in-register Cholesky instead of an SVD pseudo-inverse
//     (identical up to fp64 roundoff whenever A is non-singular, which the diagonal clamp + multiplicative
//     damping guarantee).
// One 256-thread workgroup (4 waves) per problem: with N <= 256 every thread owns one point and keeps its
// observation, world point and information matrix in registers for the whole solve (nothing is re-read).
// The whole <=10-step LM loop (with the inner reject/damp loop) runs on the device: one launch per batch of
// problems, no host round trips.  Latency-bound for a single problem (report us/solve), throughput-bound
// for large batches (report solves/s).
//
// Round 4 — the LEAN build pass (one point per thread).  TrustRegion.update measures a step against the unweighted model,
// quality = (last - loss) / -((J D)^T (2 R + J D)).  Round 3 carried the reduced pair A_u = J^T J (21) and g_u = J^T R (6) through
// every build so that any D could be scored — 27 of the 55 values of every reduction, 81 of the point's MACs, and a 42-MAC
// dependent chain per trial.  But a step's FIRST trial (the only one, in all steps but a solve's last) can be scored the way the
// reference does it: each thread keeps its own J, r from the build and forms (J_i D)^T (2 r_i + J_i D) once D exists, one more
// value in the reduction that sums the trial loss anyway.  The unweighted pair is formed only when a step's first trial has been
// REJECTED (from the same J, r: one 27-value reduction, once per solve in practice); the later trials of the step — sequential or
// in speculative rounds — score against it as before.  A build reduces 28 values instead of 55.
// Problems with more points than threads (and the one-wave throughput variant's > 64) keep the full 55-value build.
#include "common.h"
#include "pgo_math.h"
#include "pose_apply_dev.h"
#include "obs_filter_dev.h"
#include <stdlib.h>

namespace {

using namespace pgo;

// -DMV_PGO_STAMPS (profiles/probes/pgo_stamps.py builds its own copy of the library with it): s_memtime stamps of problem 0's first
// 16 LM steps, 8 per step — build start | accumulate | reduction | clamp + damp + Cholesky | SE3 update | trial residual |
// its reduction | trust region + accept — read back with mv_pgo_probe_stamps.
#ifdef MV_PGO_STAMPS
__device__ long long g_pgo_stamps[16 * 8];
#define PGO_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && steps < 16) g_pgo_stamps[steps * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
// ... and of the first two speculative reject rounds (rows 14, 15; wave 3 = the wave with the longest replay): round start | replay |
// Cholesky | SE3 update | four 64-point loss passes | quality + trust region + hand-over | walk
#define PGO_RSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 192 && dbg_rounds >= 1 && dbg_rounds <= 2) g_pgo_stamps[(13 + dbg_rounds) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PGO_STAMP(i) ((void)0)
#define PGO_RSTAMP(i) ((void)0)
#endif

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

// workgroup sum of `N` per-thread values: DPP inside the wave, LDS table across the NW waves; every thread gets all sums.
// NW = 1 (throughput variant, one wave per problem) needs no LDS and no barrier at all.  For the handful of values of a trial.
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

// The many-value reduction of a build pass (round 3).  Only the three cheapest DPP stages stay in registers (quad_perm x2 +
// row_half_mirror: every 8-lane group then holds its sum — fp64 has no DPP-fused add, and a readlane -> SGPR -> VALU round trip per
// value is a long dependent instruction stream for the single wave of a SIMD); the 8 partials per wave and value go to LDS (one
// ds_write_b64 per value with 8 lanes active), N threads add the 32 partials of one value each (four independent accumulators),
// and every thread reads the N sums back as broadcast reads.  Three barriers.
template <int N>
__device__ __forceinline__ void block_sum_wide(double* __restrict__ v /* [N] registers */, double* __restrict__ part /* [N][32] */,
                                               double* __restrict__ fin /* [N + 1] */) {
    constexpr int NW = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, t = threadIdx.x;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double s = v[k];
        s = dpp_add<0xB1>(s);    // quad_perm [1,0,3,2]
        s = dpp_add<0x4E>(s);    // quad_perm [2,3,0,1]
        s = dpp_add<0x141>(s);   // row_half_mirror: the other quad of the 8-lane half
        if ((lane & 7) == 0) part[k * (NW * 8) + wave * 8 + (lane >> 3)] = s;
    }
    __syncthreads();
    if (t < N) {
        const double* p = part + t * (NW * 8);
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int i = 0; i < NW * 8; i += 4) { a0 += p[i]; a1 += p[i + 1]; a2 += p[i + 2]; a3 += p[i + 3]; }
        fin[t] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = fin[k];
    __syncthreads();   // `part` / `fin` are rewritten by the next pass
}

// The lean build's reduction (round 4, second half).  In-kernel stamps of the form above on 28 values: 4.8 k cycles of a 13.1 k-cycle LM
// step — three dependent DPP stages per value (2 v_mov_dpp + v_add_f64 each, with their wait states) are ~420 dependent instructions
// for the single wave of a SIMD.  With at most 32 values there are enough threads to turn the job around: every thread stores its
// N values (row k = value k of all 256 threads, rows padded by 64 B so that the 8 rows a wave reads from start 16 banks apart),
// thread (k, p) = 8 k + p adds the 32 entries p, p + 8, p + 16, .. of row k (four independent accumulators), the 8 partial sums of a
// row sit in one 8-lane group and take ONE three-stage DPP sum, lane p = 0 writes the total; every thread reads the N totals back.
constexpr int RED_ROW = 256 + 8;    // doubles per row
constexpr int RED_ROWS = NLEAN + 1;   // the lean build + the quality term of a fused first trial (>= NUNW)
template <int N>
__device__ __forceinline__ void block_sum_lds(double* __restrict__ v /* [N] registers */, double* __restrict__ buf /* [N][RED_ROW] */,
                                              double* __restrict__ fin /* [N] */) {
    static_assert(N <= RED_ROWS && N * 8 <= 256, "one thread per (value, eighth)");
    const int t = threadIdx.x;
#pragma unroll
    for (int k = 0; k < N; ++k) buf[k * RED_ROW + t] = v[k];
    __syncthreads();
    if (t < N * 8) {
        const double* row = buf + (t >> 3) * RED_ROW + (t & 7);
        double a0 = row[0], a1 = row[8], a2 = row[16], a3 = row[24];
#pragma unroll
        for (int i = 4; i < 32; i += 4) { a0 += row[8 * i]; a1 += row[8 * (i + 1)]; a2 += row[8 * (i + 2)]; a3 += row[8 * (i + 3)]; }
        double s = (a0 + a1) + (a2 + a3);
        s = dpp_add<0xB1>(s);    // quad_perm [1,0,3,2]
        s = dpp_add<0x4E>(s);    // quad_perm [2,3,0,1]
        s = dpp_add<0x141>(s);   // row_half_mirror: the other quad of the 8-lane group (whole groups are active: N * 8 threads)
        if ((t & 7) == 0) fin[t >> 3] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = fin[k];
    __syncthreads();   // `buf` / `fin` are rewritten by the next pass
}

// LEAN: the one-point-per-thread passes (28 / 27 values) of the 4-wave variant go through block_sum_lds
template <int N, int NW, bool LEAN>
__device__ __forceinline__ void reduce_many(double* __restrict__ v, double* __restrict__ part, double* __restrict__ fin,
                                            double* __restrict__ lds_rows) {
    if constexpr (NW == 4 && LEAN) {
        block_sum_lds<N>(v, lds_rows, fin);
    } else if constexpr (NW == 4) {
        block_sum_wide<N>(v, part, fin);
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = wave_sum_dpp(v[k]);
    }
}

template <int GT, int NW>
__global__ __launch_bounds__(64 * NW) void pgo_solve_kernel(PgoArgs a, mvLMParams lm) {
    constexpr int PGO_THREADS = 64 * NW;
    __shared__ double red_tab[NW][NRED];
    __shared__ __attribute__((aligned(16))) double red_fin[NW == 4 ? NRED + 1 : 1];
    __shared__ __attribute__((aligned(16))) double red_rows[NW == 4 ? RED_ROWS * RED_ROW : 1];   // 59 KB (75 KB in all: two workgroups per CU still fit)
    double* const red_part = red_rows;   // the 55-value form's [NRED][32] partials: a workgroup runs one form or the other
    static_assert(NRED * 32 <= RED_ROWS * RED_ROW, "red_part aliases red_rows");
    // speculative reject rounds (NW == 4, one point per thread): every point's position / observation for the trial-loss passes, and
    // what each wave found for its trial
    constexpr int SPEC = (NW == 4) ? 1 : 0;
    __shared__ double pt_tab[SPEC ? 6 : 1][SPEC ? 64 * NW : 1];
    __shared__ int pt_valid[SPEC ? 64 * NW : 1];
    __shared__ double spec_res[SPEC ? NW : 1][13];   // per wave: ok, loss, quality, pose t[3] q[4], then the trust region BEHIND its trial: damping, tr_down, branch
    const int prob = blockIdx.x;
    const int tid = threadIdx.x;
    const int beg = a.offsets[prob], end = a.offsets[prob + 1];
    const int npts = end - beg;
    const bool cached = npts <= PGO_THREADS;  // every thread owns (at most) one point for the whole solve

    Geometry g;
    g.fx = (double)a.intrinsics[4 * prob]; g.fy = (double)a.intrinsics[4 * prob + 1];
    g.cx = (double)a.intrinsics[4 * prob + 2]; g.cy = (double)a.intrinsics[4 * prob + 3];
    g.blfx = g.fx * (double)a.baseline[prob];  // K[0,0] * bl in fp64 of the fp32 buffers

    const int live_rows = a.live_dev ? a.live_dev[(size_t)prob * a.live_stride] : a.apply_live[prob < MV_MAX_LANES ? prob : 0];
    if (a.valid_out) {      // mv_pgo_solve_posed: the lane's observation filters (obs_filter_kernel's body; PGO_THREADS == 256), then ...
        // (rows addressed through the SAME offsets table the pose-apply and the solve use — ADVICE r5: a caller with compact offsets used to get the filters of
        // rows prob * cap; the value table's row stride is the total row count offsets[nprob])
        obs_filter_rows(a.filter_inbound, a.apply_cov_Tc, a.obs2_covTc, a.filter_vals, a.filter_flags, a.filter_min_depth, a.filter_max_depth, (size_t)beg,
                        npts, live_rows < npts ? live_rows : npts, (size_t)a.offsets[gridDim.x], a.valid_out, a.count_out + prob);
        __threadfence_block();
        __syncthreads();
    }
    if (a.apply_pos_Tc) {   // ... pose_apply_kernel's rows of this problem; the solve reads both back (same workgroup)
        const float* pose = a.init_pose + 7 * prob;
        double R[9];
        mv_pose_rotation(pose, R);
        if (tid == 0 && a.apply_rot) {
#pragma unroll
            for (int i = 0; i < 9; ++i) a.apply_rot[9 * prob + i] = R[i];
        }
        const int live = live_rows;
        for (int i = tid; i < live; i += PGO_THREADS) mv_pose_apply_row(pose, R, beg + i, a.apply_pos_Tc, a.apply_cov_Tc, a.apply_pos_Tw, a.apply_cov_Tw);
        __threadfence_block();
        __syncthreads();
    }

    Pose P;
#pragma unroll
    for (int k = 0; k < 3; ++k) P.t[k] = (double)a.init_pose[7 * prob + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) P.q[k] = (double)a.init_pose[7 * prob + 3 + k];
    pose_finish(P);

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

    // acc: [0, 21) A_w, [21, 27) g_w, then  lean: [27] loss            (A_u, g_u written to [27, 54) on a step's first rejection)
    //                                        full: [27, 48) A_u, [48, 54) g_u, [54] loss
    double acc[NRED];
    PointLin lin;
    bool have_build = false;   // acc / lin already describe P: the previous step's accepted first trial was built there (MV_PGO_FUSED_BUILD)
    while (continual) {
        // ------------------------------------------------------------------ build pass
        bool have_unw = !cached;     // A_u / g_u are in acc
        double loss_build = 0.0;
        bool built_next = false;
        PGO_STAMP(0);
        if (have_build) {
            PGO_STAMP(1);   // (the previous step's accepted trial was this build)
        } else if (cached) {
            if (mine.valid) {
                accumulate_point<GT, false>(g, lm, P, mine, acc, lin);
            } else {
#pragma unroll
                for (int k = 0; k < NLEAN; ++k) acc[k] = 0.0;
            }
            PGO_STAMP(1);
            reduce_many<NLEAN, NW, true>(acc, red_part, red_fin, red_rows);
            loss_build = acc[NLEAN - 1];
            asm volatile("" : "+v"(loss_build));   // (keeps hipcc from merging the two branches' loads into one load through a pointer phi: that put acc[27] and acc[54] into scratch memory)
        } else {
#pragma unroll
            for (int k = 0; k < NRED; ++k) acc[k] = 0.0;
            for (int i = beg + tid; i < end; i += PGO_THREADS) {
                PointData<GT> d;
                load_point<GT>(a, g, lm, i, true, d);
                if (d.valid) accumulate_point<GT, true>(g, lm, P, d, acc, lin);
            }
            reduce_many<NRED, NW, false>(acc, red_part, red_fin, red_rows);
            loss_build = acc[NRED - 1];
            asm volatile("" : "+v"(loss_build));
        }
        double* Aw = acc;
        const double* gw = acc + 21;
        double* Au = acc + 27;
        const double* gu = acc + 48;

        PGO_STAMP(2);
        if (!have_loss) { loss = loss_build; loss0 = loss_build; have_loss = true; }
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
                for (int j = 0; j < 6; ++j) Aw[tri(j, j)] = fma(Aw[tri(j, j)], damping, Aw[tri(j, j)]);
                double dg6[6], D[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) dg6[j] = Aw[tri(j, j)];
                if (!chol_solve6(Aw, dg6, gw, D)) break;  // "Linear solver failed. Breaking optimization step..."

                if (reject_count == 0) PGO_STAMP(3);
                const double tp[3] = {P.t[0], P.t[1], P.t[2]}, qp[4] = {P.q[0], P.q[1], P.q[2], P.q[3]};
                se3_left_update(P, D);
                if (reject_count == 0) PGO_STAMP(4);

                // loss at the trial pose (RobustModel.loss: unweighted, uncorrected) — and, for a step's first trial of the lean
                // form, this point's term of the quality denominator
                double quality;
                if (MV_PGO_FUSED_BUILD && cached && !have_unw) {
                    // a step's first trial, lean form: the BUILD at the trial pose.  Its 28th value is the trial loss, the 29th this point's
                    // term of the quality denominator (from the J, r the previous build kept); if the trial is accepted — every step but a
                    // solve's last, in practice — these sums and `nlin` already are the next step's build pass.
                    // `lin` is overwritten in place (two PointLin per thread next to both sets of sums spilled registers): its last use on
                    // the accepted path is the quality term; a REJECTED first trial recomputes it at the restored pose (same bits, once per solve).
                    double nacc[NLEAN + 1];
                    if (mine.valid) {
                        nacc[NLEAN] = quality_point<GT>(lin, D);
                        accumulate_point<GT, false>(g, lm, P, mine, nacc, lin);
                    } else {
#pragma unroll
                        for (int k = 0; k < NLEAN + 1; ++k) nacc[k] = 0.0;
                    }
                    PGO_STAMP(5);
                    reduce_many<NLEAN + 1, NW, true>(nacc, red_part, red_fin, red_rows);
                    PGO_STAMP(6);
                    loss = nacc[NLEAN - 1];
                    quality = (last - loss) / -nacc[NLEAN];
                    if (!(last < loss && reject_count < lm.reject)) {   // accepted (decided again below, on the same values): keep the build
#pragma unroll
                        for (int k = 0; k < NLEAN; ++k) acc[k] = nacc[k];
                        built_next = true;
                    }
                } else if (cached) {
                    double lq[2] = {0.0, 0.0};
                    if (mine.valid) {
                        lq[0] = point_loss<GT>(g, lm, P, mine.pw, mine.obs);
                        if (!have_unw) lq[1] = quality_point<GT>(lin, D);
                    }
                    if (reject_count == 0) PGO_STAMP(5);
                    block_sum<2, NW>(lq, red_tab);
                    if (reject_count == 0) PGO_STAMP(6);
                    loss = lq[0];
                    if (have_unw) {
                        double Dq[6];   // (opaque for the same reason: the 42-MAC quadratic form was evaluated — and discarded — in every step)
#pragma unroll
                        for (int j = 0; j < 6; ++j) { Dq[j] = D[j]; asm volatile("" : "+v"(Dq[j])); }
                        quality = tr_quality(Dq, gu, Au, last, loss);
                    } else {
                        quality = (last - loss) / -lq[1];
                    }
                } else {
                    double la[1] = {0.0};
                    for (int i = beg + tid; i < end; i += PGO_THREADS) {
                        PointData<GT> d;
                        load_point<GT>(a, g, lm, i, true, d);
                        if (d.valid) la[0] += point_loss<GT>(g, lm, P, d.pw, d.obs);
                    }
                    block_sum<1, NW>(la, red_tab);
                    loss = la[0];
                    quality = tr_quality(D, gu, Au, last, loss);
                }
                pred_branch = tr_update(lm, quality, damping, tr_down);

                if (last < loss && reject_count < lm.reject) {  // reject step
                    P.t[0] = tp[0]; P.t[1] = tp[1]; P.t[2] = tp[2];
                    P.q[0] = qp[0]; P.q[1] = qp[1]; P.q[2] = qp[2]; P.q[3] = qp[3];
                    pose_finish(P);
                    loss = last;
                    reject_count += 1;
                    if (!have_unw) {   // the step's later trials are scored against the reduced unweighted pair
                        double u[NUNW];
                        if (MV_PGO_FUSED_BUILD && cached && mine.valid) {   // `lin` holds the rejected trial pose's J, r: back to the restored pose's
                            double scratch_acc[NLEAN];
                            accumulate_point<GT, false>(g, lm, P, mine, scratch_acc, lin);
                        }
                        if (mine.valid) {
                            // (opaque: without it hipcc SPECULATES these 99 fp64 operations — they have no side effects — into every LM
                            // step in front of the Cholesky and selects the results away: seen in the ISA and in the stamps, 1.4 k cycles per step)
                            asm volatile("" : "+v"(lin.s2));
                            unweighted_point<GT>(lin, u);
                        } else {
#pragma unroll
                            for (int k = 0; k < NUNW; ++k) u[k] = 0.0;
                        }
                        reduce_many<NUNW, NW, true>(u, red_part, red_fin, red_rows);
#pragma unroll
                        for (int k = 0; k < NUNW; ++k) Au[k] = u[k];
                        have_unw = true;
                    }
                } else {
                    break;
                }
            } else {
                const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
                PGO_RSTAMP(0);
                // ---- this wave's trial: the (wv + 1)-th from here, assuming the wv before it are rejected through the predicted branch
                double dg6[6], damp_s = damping, trd_s = tr_down;
#pragma unroll
                for (int j = 0; j < 6; ++j) dg6[j] = Aw[tri(j, j)];
                for (int i = 0; i <= wv; ++i) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) dg6[j] = fma(dg6[j], damp_s, dg6[j]);
                    if (i < wv) tr_apply(lm, pred_branch, damp_s, trd_s);
                }
                PGO_RSTAMP(1);
                double D[6];
                const bool ok = chol_solve6(Aw, dg6, gw, D);
                PGO_RSTAMP(2);
                Pose Pw = P;
                double loss_w = 0.0, quality_w = 0.0;
                if (ok) {
                    se3_left_update(Pw, D);
                    PGO_RSTAMP(3);
                    double part[NW];
#pragma unroll
                    for (int c = 0; c < NW; ++c) {
                        const int i = c * 64 + lane;
                        double v = 0.0;
                        if (pt_valid[i]) {
                            const double pw[3] = {pt_tab[0][i], pt_tab[1][i], pt_tab[2][i]};
                            const double ob[3] = {pt_tab[3][i], pt_tab[4][i], pt_tab[5][i]};
                            v = point_loss<GT>(g, lm, Pw, pw, ob);
                        }
                        part[c] = wave_sum_dpp(v);
                    }
                    loss_w = part[0];
#pragma unroll
                    for (int c = 1; c < NW; ++c) loss_w += part[c];
                    PGO_RSTAMP(4);
                    quality_w = tr_quality(D, gu, Au, last, loss_w);
                }
                // the trust-region update BEHIND this trial, from the replayed state: it IS the true one whenever the walk below gets as
                // far as this trial (every earlier trial of the round rejected through the predicted branch), so the walk reads it instead
                // of redoing four dependent updates (two fp64 divisions each) one after the other
                int branch_w = 0;
                if (ok) branch_w = tr_update(lm, quality_w, damp_s, trd_s);
                if (lane == 0) {
                    double* o = spec_res[wv];
                    o[0] = ok ? 1.0 : 0.0; o[1] = loss_w; o[2] = quality_w;
                    o[3] = Pw.t[0]; o[4] = Pw.t[1]; o[5] = Pw.t[2];
                    o[6] = Pw.q[0]; o[7] = Pw.q[1]; o[8] = Pw.q[2]; o[9] = Pw.q[3];
                    o[10] = damp_s; o[11] = trd_s; o[12] = (double)branch_w;
                }
                __syncthreads();
                PGO_RSTAMP(5);
                // ---- the sequential form's bookkeeping over the four results
                bool leave = false;
                double w_ok[NW], w_loss[NW], w_damp[NW], w_trd[NW], w_br[NW];   // all four results up front: one LDS round trip, not one per trial
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    const double* o = spec_res[i];
                    w_ok[i] = o[0]; w_loss[i] = o[1]; w_damp[i] = o[10]; w_trd[i] = o[11]; w_br[i] = o[12];
                }
#pragma unroll
                for (int i = 0; i < NW; ++i) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) Aw[tri(j, j)] = fma(Aw[tri(j, j)], damping, Aw[tri(j, j)]);
                    if (w_ok[i] == 0.0) { leave = true; break; }       // "Linear solver failed"
                    loss = w_loss[i];
                    damping = w_damp[i]; tr_down = w_trd[i];
                    const int branch = (int)w_br[i];
                    if (last < loss && reject_count < lm.reject) {   // reject step
                        loss = last;
                        reject_count += 1;
                        if (branch != pred_branch) { pred_branch = branch; break; }   // the later trials of this round started from other inputs
                    } else {
                        const double* o = spec_res[i];
                        P.t[0] = o[3]; P.t[1] = o[4]; P.t[2] = o[5];
                        P.q[0] = o[6]; P.q[1] = o[7]; P.q[2] = o[8]; P.q[3] = o[9];
                        pose_finish(P);
                        leave = true;
                        break;
                    }
                }
                __syncthreads();   // spec_res is rewritten by the next round
                PGO_RSTAMP(6);
                if (leave) break;
            }
        }

        // ------------------------------------------------------------------ StopOnPlateau.step(loss)
        PGO_STAMP(7);
        steps += 1;
        if (steps >= lm.max_steps) continual = false;
        if ((last - loss) < lm.decreasing) patience_count += 1; else patience_count = 0;
        if (patience_count >= lm.patience) continual = false;
        if (lm.stop_on_reject > 0 && reject_count >= lm.stop_on_reject) continual = false;
        have_build = built_next;
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
        if (a.pose_sink) {
            float* of = a.pose_sink + 7 * (size_t)prob;
            of[0] = (float)P.t[0]; of[1] = (float)P.t[1]; of[2] = (float)P.t[2];
            of[3] = (float)P.q[0]; of[4] = (float)P.q[1]; of[5] = (float)P.q[2]; of[6] = (float)P.q[3];
        }
    }
}

}  // namespace

#ifdef MV_PGO_STAMPS
extern "C" int mv_pgo_probe_stamps(long long* host_out /* [128] */) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_pgo_stamps), sizeof(long long) * 128) == hipSuccess ? MV_OK : MV_ERR_LAUNCH;
}
#endif

extern "C" void mv_lm_default_params(mvLMParams* p) {
    if (!p) return;
    p->huber_delta = 0.1;
    p->radius = 1e3;
    p->tr_high = 0.5; p->tr_low = 1e-3; p->tr_up = 2.0; p->tr_down = 0.5; p->tr_factor = 0.5;
    p->tr_min = 1e-6; p->tr_max = 1e16;
    p->diag_min = 1e-6; p->diag_max = 1e32;
    p->decreasing = 1e-5;
    p->pinv_rcond = 1e-15;
    p->reject = 16; p->max_steps = 10; p->patience = 2; p->stop_on_reject = 1;
}

struct PoseApplyArgs {   // mv_pgo_solve_posed's extra arguments (all null for the plain solve)
    const float* pos_Tc;
    const double* cov_Tc;
    double* out_rot;
    float* pose_sink;
    const int32_t* n_live;
    int filter_flags = -1;
    const int32_t* live_dev = nullptr;
    int live_stride = 0;
    float filter_min_depth = 0.f, filter_max_depth = 0.f;
    int cap = 0;
    const uint8_t* inbound = nullptr;
    const float* vals = nullptr;
    uint8_t* valid_out = nullptr;
    int32_t* count_out = nullptr;
};

static int pgo_solve_impl(int nprob, const int32_t* offsets, int graph_type, const float* init_pose,
                          const float* intrinsics, const float* baseline, const float* pos_Tw, const double* cov_Tw,
                          const float* pixel2_uv, const float* pixel2_d, const float* pixel2_disp,
                          const float* pixel2_disp_cov, const float* pixel2_uv_cov, const double* obs2_covTc,
                          const uint8_t* valid, int min_points, const mvLMParams* params, double* out_pose,
                          double* out_info, float* out_pose_f32, const PoseApplyArgs& pa, mvStream_t stream) {
    MV_CHECK_ARG(nprob >= 0 && params);
    if (nprob == 0) return MV_OK;
    MV_CHECK_ARG(offsets && init_pose && intrinsics && baseline && pos_Tw && pixel2_uv && out_pose && out_info);
    MV_CHECK_ARG(params->max_steps >= 1 && params->reject >= 0 && params->stop_on_reject >= 0 && params->radius > 0 && params->huber_delta > 0);
    PgoArgs a{offsets, init_pose, intrinsics, baseline, pos_Tw, cov_Tw, pixel2_uv, pixel2_d, pixel2_disp,
              pixel2_disp_cov, pixel2_uv_cov, obs2_covTc, valid, min_points, out_pose, out_info, out_pose_f32, 1};
    a.pose_sink = pa.pose_sink;
    if (pa.pos_Tc) {
        MV_CHECK_ARG(nprob <= MV_MAX_LANES && (pa.n_live || pa.live_dev));
        a.live_dev = pa.live_dev;
        a.live_stride = pa.live_stride;
        a.apply_pos_Tc = pa.pos_Tc;
        a.apply_cov_Tc = pa.cov_Tc;
        a.apply_pos_Tw = const_cast<float*>(pos_Tw);        // the solve's own input tables are the outputs of the fold
        a.apply_cov_Tw = pa.cov_Tc ? const_cast<double*>(cov_Tw) : nullptr;
        a.apply_rot = pa.out_rot;
        for (int l = 0; l < nprob && pa.n_live; ++l) {
            MV_CHECK_ARG(pa.n_live[l] >= 0);
            a.apply_live[l] = pa.n_live[l];
        }
        if (pa.filter_flags >= 0) {
            MV_CHECK_ARG(pa.valid_out && pa.count_out && pa.cap >= 0 && (!(pa.filter_flags & 1) || (pa.cov_Tc && obs2_covTc)) &&
                         (!(pa.filter_flags & 6) || pa.vals));
            a.filter_flags = pa.filter_flags;
            a.filter_min_depth = pa.filter_min_depth;
            a.filter_max_depth = pa.filter_max_depth;
            a.filter_cap = pa.cap;
            a.filter_inbound = pa.inbound;
            a.filter_vals = pa.vals;
            a.valid_out = pa.valid_out;
            a.count_out = pa.count_out;
            a.valid = pa.valid_out;      // what the solve reads
            for (int l = 0; l < nprob && pa.n_live; ++l) MV_CHECK_ARG(pa.n_live[l] <= pa.cap);
        }
    }
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
    if (wide) hipLaunchKernelGGL((pgo_solve_kernel<G, 4>), grid, block, 0, s, a, *params);    \
    else hipLaunchKernelGGL((pgo_solve_kernel<G, 1>), grid, block, 0, s, a, *params)
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

extern "C" int mv_pgo_solve(int nprob, const int32_t* offsets, int graph_type, const float* init_pose,
                            const float* intrinsics, const float* baseline, const float* pos_Tw, const double* cov_Tw,
                            const float* pixel2_uv, const float* pixel2_d, const float* pixel2_disp,
                            const float* pixel2_disp_cov, const float* pixel2_uv_cov, const double* obs2_covTc,
                            const uint8_t* valid, int min_points, const mvLMParams* params, double* out_pose,
                            double* out_info, float* out_pose_f32, mvStream_t stream) {
    return pgo_solve_impl(nprob, offsets, graph_type, init_pose, intrinsics, baseline, pos_Tw, cov_Tw, pixel2_uv, pixel2_d, pixel2_disp,
                          pixel2_disp_cov, pixel2_uv_cov, obs2_covTc, valid, min_points, params, out_pose, out_info, out_pose_f32,
                          PoseApplyArgs{nullptr, nullptr, nullptr, nullptr, nullptr}, stream);
}

extern "C" int mv_pgo_solve_posed(int nprob, const int32_t* offsets, const int32_t* n_live, int cap, int graph_type, const float* init_pose,
                                  const float* intrinsics, const float* baseline, const float* pos_Tc, const double* cov_Tc,
                                  float* pos_Tw, double* cov_Tw, double* out_rot, const float* pixel2_uv, const float* pixel2_d,
                                  const float* pixel2_disp, const float* pixel2_disp_cov, const float* pixel2_uv_cov,
                                  const double* obs2_covTc, int filter_flags, float filter_min_depth, float filter_max_depth,
                                  const uint8_t* inbound, const float* vals, uint8_t* valid, int32_t* count_out, int min_points,
                                  const mvLMParams* params, double* out_pose, double* out_info, float* out_pose_f32, float* pose_sink,
                                  mvStream_t stream) {
    MV_CHECK_ARG(pos_Tc && pos_Tw && n_live && (!cov_Tc || cov_Tw));
    MV_CHECK_ARG(nprob < 512);   // (the 256-thread solve variant: the filter body is written for it)
    PoseApplyArgs pa{pos_Tc, cov_Tc, out_rot, pose_sink, n_live};
    if (filter_flags >= 0) {
        pa.filter_flags = filter_flags; pa.filter_min_depth = filter_min_depth; pa.filter_max_depth = filter_max_depth; pa.cap = cap;
        pa.inbound = inbound; pa.vals = vals; pa.valid_out = valid; pa.count_out = count_out;
    }
    return pgo_solve_impl(nprob, offsets, graph_type, init_pose, intrinsics, baseline, pos_Tw, cov_Tw, pixel2_uv, pixel2_d, pixel2_disp,
                          pixel2_disp_cov, pixel2_uv_cov, obs2_covTc, valid, min_points, params, out_pose, out_info, out_pose_f32, pa, stream);
}

// mv_pgo_solve_posed of the device-driven frame (round 6): the live-row count of problem l is read from device memory (n_live_dev[l * n_live_stride] <= cap,
// written by mv_backend_front_draw_lanes earlier on the stream) — the host never learns it.
extern "C" int mv_pgo_solve_posed_dev(int nprob, const int32_t* offsets, const int32_t* n_live_dev, int n_live_stride, int cap, int graph_type,
                                      const float* init_pose, const float* intrinsics, const float* baseline, const float* pos_Tc, const double* cov_Tc,
                                      float* pos_Tw, double* cov_Tw, double* out_rot, const float* pixel2_uv, const float* pixel2_d,
                                      const float* pixel2_disp, const float* pixel2_disp_cov, const float* pixel2_uv_cov, const double* obs2_covTc,
                                      int filter_flags, float filter_min_depth, float filter_max_depth, const uint8_t* inbound, const float* vals,
                                      uint8_t* valid, int32_t* count_out, int min_points, const mvLMParams* params, double* out_pose, double* out_info,
                                      float* out_pose_f32, float* pose_sink, mvStream_t stream) {
    MV_CHECK_ARG(pos_Tc && pos_Tw && n_live_dev && n_live_stride >= 1 && (!cov_Tc || cov_Tw));
    MV_CHECK_ARG(nprob < 512);
    PoseApplyArgs pa{pos_Tc, cov_Tc, out_rot, pose_sink, nullptr};
    pa.live_dev = n_live_dev;
    pa.live_stride = n_live_stride;
    if (filter_flags >= 0) {
        pa.filter_flags = filter_flags; pa.filter_min_depth = filter_min_depth; pa.filter_max_depth = filter_max_depth; pa.cap = cap;
        pa.inbound = inbound; pa.vals = vals; pa.valid_out = valid; pa.count_out = count_out;
    }
    return pgo_solve_impl(nprob, offsets, graph_type, init_pose, intrinsics, baseline, pos_Tw, cov_Tw, pixel2_uv, pixel2_d, pixel2_disp,
                          pixel2_disp_cov, pixel2_uv_cov, obs2_covTc, valid, min_points, params, out_pose, out_info, out_pose_f32, pa, stream);
}
