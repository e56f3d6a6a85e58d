// TEST INFRASTRUCTURE — a host replay of mac-vo_amd/csrc/pgo_solve.hip, lane by lane.
//
// The kernel's per-point / per-pose arithmetic is mac-vo_amd/csrc/pgo_math.h; this file includes that very header (g++,
// -ffp-contract=off, fma() = the hardware / libm fused multiply-add) and replays the kernel's control flow around it: 64 * nw
// "threads" per problem, the same ownership of points, the same reduction trees in the same order (DPP butterfly inside a wave,
// LDS tables across waves), the LM loop with its sequential and speculative reject rounds.  The only arithmetic that differs from
// the device is rsqrt (1 / sqrt here, v_rsq_f64 + Newton steps there).  The CPU suite checks it against the oracle (oracle/pgo.py)
// and the reference golden (tests/golden/pgo.npz: pose, LM steps, reject count, loss), so a change to the kernel's arithmetic or
// loop structure is validated before it reaches a GPU; the GPU suite then checks kernel == twin to roundoff.
// Nothing in the product path builds, links or loads this file.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../mac-vo_amd/csrc/pgo_math.h"

using namespace pgo;

namespace {

// ---- the kernel's reduction trees -------------------------------------------------------------------------
// three DPP stages: every 8-lane group holds its sum (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror)
void dpp3(const double* v, double* s3) {
    double s1[64], s2[64];
    for (int i = 0; i < 64; ++i) s1[i] = v[i] + v[i ^ 1];
    for (int i = 0; i < 64; ++i) s2[i] = s1[i] + s1[i ^ 2];
    for (int i = 0; i < 64; ++i) s3[i] = s2[i] + s2[(i & ~7) | (7 - (i & 7))];
}
double wave_sum_dpp(const double* v) {
    double s3[64], s4[64];
    dpp3(v, s3);
    for (int i = 0; i < 64; ++i) s4[i] = s3[i] + s3[(i & ~15) | (15 - (i & 15))];   // row_mirror
    return (s4[0] + s4[16]) + (s4[32] + s4[48]);
}
// block_sum<1 value, NW>: per-wave sums added in wave order
double block_sum1(const double* v /* [64 nw] */, int nw) {
    double s = wave_sum_dpp(v);
    for (int w = 1; w < nw; ++w) s += wave_sum_dpp(v + 64 * w);
    return s;
}
// reduce_many: nw == 1 -> DPP only; nw == 4, lean (one point per thread) -> block_sum_lds: thread (k, p) adds entries p, p + 8, .. of
// the 256 with four accumulators, the 8 partial sums take the three-stage DPP tree; nw == 4 otherwise -> block_sum_wide (8-lane DPP
// partials, 32 of them added by one thread with four accumulators)
double reduce_many1(const double* v, int nw, bool lean) {
    if (nw != 4) return wave_sum_dpp(v);
    if (lean) {
        double s[8];
        for (int p = 0; p < 8; ++p) {
            double a0 = v[p], a1 = v[p + 8], a2 = v[p + 16], a3 = v[p + 24];
            for (int i = 4; i < 32; i += 4) { a0 += v[p + 8 * i]; a1 += v[p + 8 * (i + 1)]; a2 += v[p + 8 * (i + 2)]; a3 += v[p + 8 * (i + 3)]; }
            s[p] = (a0 + a1) + (a2 + a3);
        }
        double x1[8], x2[8];
        for (int i = 0; i < 8; ++i) x1[i] = s[i] + s[i ^ 1];
        for (int i = 0; i < 8; ++i) x2[i] = x1[i] + x1[i ^ 2];
        return x2[0] + x2[7];
    }
    double part[32];
    for (int w = 0; w < 4; ++w) {
        double s3[64];
        dpp3(v + 64 * w, s3);
        for (int l = 0; l < 64; l += 8) part[w * 8 + (l >> 3)] = s3[l];
    }
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int i = 0; i < 32; i += 4) { a0 += part[i]; a1 += part[i + 1]; a2 += part[i + 2]; a3 += part[i + 3]; }
    return (a0 + a1) + (a2 + a3);
}

template <int GT>
void solve_one(const PgoArgs& a, const mvLMParams& lm, int prob, int nw) {
    const int T = 64 * nw;
    const bool SPEC = nw == 4;
    const int beg = a.offsets[prob], end = a.offsets[prob + 1];
    const int npts = end - beg;
    const bool cached = npts <= T;

    Geometry g;
    g.fx = (double)a.intrinsics[4 * prob]; g.fy = (double)a.intrinsics[4 * prob + 1];
    g.cx = (double)a.intrinsics[4 * prob + 2]; g.cy = (double)a.intrinsics[4 * prob + 3];
    g.blfx = g.fx * (double)a.baseline[prob];

    Pose P;
    for (int k = 0; k < 3; ++k) P.t[k] = (double)a.init_pose[7 * prob + k];
    for (int k = 0; k < 4; ++k) P.q[k] = (double)a.init_pose[7 * prob + 3 + k];
    pose_finish(P);

    std::vector<PointData<GT>> mine(T);
    std::vector<PointLin> lin(T);
    for (int t = 0; t < T; ++t) {
        mine[t].valid = false;
        if (cached) load_point<GT>(a, g, lm, beg + t, t < npts, mine[t]);
    }

    double damping = 1.0 / lm.radius, tr_down = lm.tr_down;
    double loss = 0.0, last = 0.0, loss0 = 0.0;
    bool have_loss = false;
    int steps = 0, patience_count = 0, reject_count = 0;
    bool continual = true;
    int dbg_rounds = 0, dbg_trials = 0;
    int pred_branch = 3;

    std::vector<double> col(T);
    {
        for (int t = 0; t < T; ++t) {
            double nv = 0.0;
            if (cached) nv = mine[t].valid ? 1.0 : 0.0;
            else for (int i = beg + t; i < end; i += T) nv += (a.valid ? (a.valid[i] != 0) : 1) ? 1.0 : 0.0;
            col[t] = nv;
        }
        if ((int)block_sum1(col.data(), nw) < a.min_points) continual = false;
    }

    std::vector<double> per(T * NRED);
    double acc[NRED];
    bool have_build = false;
    while (continual) {
        bool have_unw = !cached;
        double loss_build = 0.0;
        bool built_next = false;
        const int n_build = cached ? NLEAN : NRED;
        for (int t = 0; t < T && !have_build; ++t) {
            double* o = &per[(size_t)t * NRED];
            for (int k = 0; k < NRED; ++k) o[k] = 0.0;
            if (cached) {
                if (mine[t].valid) accumulate_point<GT, false>(g, lm, P, mine[t], o, lin[t]);
            } else {
                for (int i = beg + t; i < end; i += T) {
                    PointData<GT> d;
                    load_point<GT>(a, g, lm, i, true, d);
                    if (d.valid) accumulate_point<GT, true>(g, lm, P, d, o, lin[t]);
                }
            }
        }
        if (!have_build) {
            for (int k = 0; k < n_build; ++k) {
                for (int t = 0; t < T; ++t) col[t] = per[(size_t)t * NRED + k];
                acc[k] = reduce_many1(col.data(), nw, cached);
            }
            loss_build = acc[n_build - 1];
        }
        double* Aw = acc;
        const double* gw = acc + 21;
        double* Au = acc + 27;
        const double* gu = acc + 48;

        if (!have_loss) { loss = loss_build; loss0 = loss_build; have_loss = true; }
        last = loss;
        reject_count = 0;
        for (int j = 0; j < 6; ++j) Aw[tri(j, j)] = fmin(fmax(Aw[tri(j, j)], lm.diag_min), lm.diag_max);

        while (last <= loss) {
            dbg_trials += 1;
            const bool spec_round = SPEC && cached && reject_count >= 1 && a.spec != 0;
            if (spec_round) dbg_rounds += 1;
            if (!spec_round) {
                for (int j = 0; j < 6; ++j) Aw[tri(j, j)] = fma(Aw[tri(j, j)], damping, Aw[tri(j, j)]);
                double dg6[6], D[6];
                for (int j = 0; j < 6; ++j) dg6[j] = Aw[tri(j, j)];
                if (!chol_solve6(Aw, dg6, gw, D)) break;
                const double tp[3] = {P.t[0], P.t[1], P.t[2]}, qp[4] = {P.q[0], P.q[1], P.q[2], P.q[3]};
                se3_left_update(P, D);
                double quality;
                if (MV_PGO_FUSED_BUILD && cached && !have_unw) {
                    std::vector<double> nper((size_t)T * (NLEAN + 1), 0.0);
                    for (int t = 0; t < T; ++t)
                        if (mine[t].valid) {
                            nper[(size_t)t * (NLEAN + 1) + NLEAN] = quality_point<GT>(lin[t], D);
                            accumulate_point<GT, false>(g, lm, P, mine[t], &nper[(size_t)t * (NLEAN + 1)], lin[t]);      // lin in place, as the kernel
                        }
                    double nacc[NLEAN + 1];
                    for (int k = 0; k < NLEAN + 1; ++k) {
                        for (int t = 0; t < T; ++t) col[t] = nper[(size_t)t * (NLEAN + 1) + k];
                        nacc[k] = reduce_many1(col.data(), nw, true);
                    }
                    loss = nacc[NLEAN - 1];
                    quality = (last - loss) / -nacc[NLEAN];
                    if (!(last < loss && reject_count < lm.reject)) {
                        for (int k = 0; k < NLEAN; ++k) acc[k] = nacc[k];
                        built_next = true;
                    }
                } else if (cached) {
                    std::vector<double> l0(T, 0.0), l1(T, 0.0);
                    for (int t = 0; t < T; ++t)
                        if (mine[t].valid) {
                            l0[t] = point_loss<GT>(g, lm, P, mine[t].pw, mine[t].obs);
                            if (!have_unw) l1[t] = quality_point<GT>(lin[t], D);
                        }
                    loss = block_sum1(l0.data(), nw);
                    const double qd = block_sum1(l1.data(), nw);
                    quality = have_unw ? tr_quality(D, gu, Au, last, loss) : (last - loss) / -qd;
                } else {
                    for (int t = 0; t < T; ++t) {
                        double la = 0.0;
                        for (int i = beg + t; i < end; i += T) {
                            PointData<GT> d;
                            load_point<GT>(a, g, lm, i, true, d);
                            if (d.valid) la += point_loss<GT>(g, lm, P, d.pw, d.obs);
                        }
                        col[t] = la;
                    }
                    loss = block_sum1(col.data(), nw);
                    quality = tr_quality(D, gu, Au, last, loss);
                }
                pred_branch = tr_update(lm, quality, damping, tr_down);
                if (last < loss && reject_count < lm.reject) {
                    for (int k = 0; k < 3; ++k) P.t[k] = tp[k];
                    for (int k = 0; k < 4; ++k) P.q[k] = qp[k];
                    pose_finish(P);
                    loss = last;
                    reject_count += 1;
                    if (!have_unw) {
                        std::vector<double> u((size_t)T * NUNW, 0.0);
                        if (MV_PGO_FUSED_BUILD && cached)
                            for (int t = 0; t < T; ++t)
                                if (mine[t].valid) { double tmp[NLEAN]; accumulate_point<GT, false>(g, lm, P, mine[t], tmp, lin[t]); }   // lin back to the restored pose's
                        for (int t = 0; t < T; ++t)
                            if (mine[t].valid) unweighted_point<GT>(lin[t], &u[(size_t)t * NUNW]);
                        for (int k = 0; k < NUNW; ++k) {
                            for (int t = 0; t < T; ++t) col[t] = u[(size_t)t * NUNW + k];
                            Au[k] = reduce_many1(col.data(), nw, true);
                        }
                        have_unw = true;
                    }
                } else {
                    break;
                }
            } else {
                double res[4][13];
                for (int wv = 0; wv < 4; ++wv) {
                    double dg6[6], damp_s = damping, trd_s = tr_down;
                    for (int j = 0; j < 6; ++j) dg6[j] = Aw[tri(j, j)];
                    for (int i = 0; i <= wv; ++i) {
                        for (int j = 0; j < 6; ++j) dg6[j] = fma(dg6[j], damp_s, dg6[j]);
                        if (i < wv) tr_apply(lm, pred_branch, damp_s, trd_s);
                    }
                    double D[6];
                    const bool ok = chol_solve6(Aw, dg6, gw, D);
                    Pose Pw = P;
                    double loss_w = 0.0, quality_w = 0.0;
                    if (ok) {
                        se3_left_update(Pw, D);
                        double part[4];
                        for (int c = 0; c < 4; ++c) {
                            double v[64];
                            for (int l = 0; l < 64; ++l) {
                                const int i = c * 64 + l;
                                v[l] = mine[i].valid ? point_loss<GT>(g, lm, Pw, mine[i].pw, mine[i].obs) : 0.0;
                            }
                            part[c] = wave_sum_dpp(v);
                        }
                        loss_w = part[0];
                        for (int c = 1; c < 4; ++c) loss_w += part[c];
                        quality_w = tr_quality(D, gu, Au, last, loss_w);
                    }
                    int branch_w = 0;
                    if (ok) branch_w = tr_update(lm, quality_w, damp_s, trd_s);
                    double* o = res[wv];
                    o[0] = ok ? 1.0 : 0.0; o[1] = loss_w; o[2] = quality_w;
                    o[3] = Pw.t[0]; o[4] = Pw.t[1]; o[5] = Pw.t[2];
                    o[6] = Pw.q[0]; o[7] = Pw.q[1]; o[8] = Pw.q[2]; o[9] = Pw.q[3];
                    o[10] = damp_s; o[11] = trd_s; o[12] = (double)branch_w;
                }
                bool leave = false;
                for (int i = 0; i < 4; ++i) {
                    for (int j = 0; j < 6; ++j) Aw[tri(j, j)] = fma(Aw[tri(j, j)], damping, Aw[tri(j, j)]);
                    const double* o = res[i];
                    if (o[0] == 0.0) { leave = true; break; }
                    loss = o[1];
                    damping = o[10]; tr_down = o[11];
                    const int branch = (int)o[12];
                    if (last < loss && reject_count < lm.reject) {
                        loss = last;
                        reject_count += 1;
                        if (branch != pred_branch) { pred_branch = branch; break; }
                    } else {
                        P.t[0] = o[3]; P.t[1] = o[4]; P.t[2] = o[5];
                        P.q[0] = o[6]; P.q[1] = o[7]; P.q[2] = o[8]; P.q[3] = o[9];
                        pose_finish(P);
                        leave = true;
                        break;
                    }
                }
                if (leave) break;
            }
        }

        steps += 1;
        if (steps >= lm.max_steps) continual = false;
        if ((last - loss) < lm.decreasing) patience_count += 1; else patience_count = 0;
        if (patience_count >= lm.patience) continual = false;
        if (lm.stop_on_reject > 0 && reject_count >= lm.stop_on_reject) continual = false;
        have_build = built_next;
    }

    double* o = a.out_pose + 7 * (size_t)prob;
    o[0] = P.t[0]; o[1] = P.t[1]; o[2] = P.t[2];
    o[3] = P.q[0]; o[4] = P.q[1]; o[5] = P.q[2]; o[6] = P.q[3];
    double* inf = a.out_info + 4 * (size_t)prob;
    inf[0] = loss; inf[1] = (double)steps; inf[2] = (double)reject_count; inf[3] = loss0;
    if (a.spec == 2) inf[3] = (double)(dbg_rounds * 1000 + dbg_trials);
    if (a.out_pose_f32) {
        float* of = a.out_pose_f32 + 7 * (size_t)prob;
        for (int k = 0; k < 7; ++k) of[k] = (float)o[k];
    }
}

}  // namespace

// same arguments as mv_pgo_solve (include/macvo_hip.h) on HOST arrays, plus nw (4: the latency variant, 1: the throughput
// variant; 0: the library's choice by nprob) and spec (MV_PGO_SPEC)
extern "C" int pgo_twin_solve(int nprob, const int32_t* offsets, int graph_type, const float* init_pose, const float* intrinsics,
                              const float* baseline, const float* pos_Tw, const double* cov_Tw, const float* pixel2_uv,
                              const float* pixel2_d, const float* pixel2_disp, const float* pixel2_disp_cov, const float* pixel2_uv_cov,
                              const double* obs2_covTc, const uint8_t* valid, int min_points, const mvLMParams* params, double* out_pose,
                              double* out_info, float* out_pose_f32, int nw, int spec) {
    if (nprob < 0 || !params) return 1;
    if (nw == 0) nw = nprob < 512 ? 4 : 1;
    if (nw != 1 && nw != 4) return 1;
    PgoArgs a{offsets, init_pose, intrinsics, baseline, pos_Tw, cov_Tw, pixel2_uv, pixel2_d, pixel2_disp,
              pixel2_disp_cov, pixel2_uv_cov, obs2_covTc, valid, min_points, out_pose, out_info, out_pose_f32, spec};
    for (int p = 0; p < nprob; ++p) {
        switch (graph_type) {
            case MV_GRAPH_ICP: solve_one<MV_GRAPH_ICP>(a, *params, p, nw); break;
            case MV_GRAPH_REPROJ: solve_one<MV_GRAPH_REPROJ>(a, *params, p, nw); break;
            case MV_GRAPH_DISP: solve_one<MV_GRAPH_DISP>(a, *params, p, nw); break;
            default: return 1;
        }
    }
    return 0;
}
