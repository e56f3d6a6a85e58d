// Device body of the observation filters (Module/OutlierFilter.py:91-141: CovarianceSanityFilter, SimpleDepthFilter, LikelyFrontOfCamFilter), shared by
// obs_filter_kernel (frontend_ops.hip) and the prologue of pgo_solve_kernel (mv_pgo_solve_posed folds the filters of a lane into its solve) so that both
// produce the same bits.
#pragma once
#include "common.h"
#include <math.h>

// the observation filters of one pipeline lane, run by one whole workgroup (obs_filter_kernel: blockIdx.x = lane; pgo_solve_kernel's prologue: the lane's solve workgroup)
__device__ __forceinline__ void obs_filter_rows(const uint8_t* inbound, const double* cov1, const double* cov2, const float* vals, int flags,
                                                float min_depth, float max_depth, size_t row0, int cap, int N, size_t vs, uint8_t* valid,
                                                int32_t* count) {
    // one workgroup per lane: N <= a few thousand observations.  The lane's rows are [row0, row0 + cap) of every table (`vals`: SoA [11, vs], vs = rows of
    // all lanes); rows in [N, cap) are written as invalid so that a capacity-strided solve (mv_pgo_solve with static offsets) skips them.
    {
        if (inbound) inbound += row0;
        if (cov1) cov1 += 9 * row0;
        if (cov2) cov2 += 9 * row0;
        if (vals) vals += row0;
        valid += row0;
    }
    __shared__ int total;
    if (threadIdx.x == 0) total = 0;
    // LikelyFrontOfCamFilter: if ANY pixel1_d_cov is the -1 placeholder the filter lets every row pass (:133-136)
    int has_placeholder = 0;
    if (flags & 4)
        for (int n = threadIdx.x; n < N; n += blockDim.x) has_placeholder |= (vals[3 * vs + n] == -1.f);
    const bool front_off = __syncthreads_or(has_placeholder) != 0;
    int local = 0;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        bool ok = inbound ? inbound[n] != 0 : true;
        if (ok && (flags & 1)) {  // CovarianceSanityFilter (OutlierFilter.py:91-100)
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const double a = cov1[(size_t)n * 9 + k], b = cov2[(size_t)n * 9 + k];
                ok = ok && isfinite(a) && isfinite(b);
            }
        }
        if (ok && (flags & 6)) {
            const float d1 = vals[n], d2 = vals[4 * vs + n], c1 = vals[3 * vs + n], c2 = vals[7 * vs + n];
            if (flags & 2)  // SimpleDepthFilter (:103-121)
                ok = !((d1 < min_depth) || (d1 > max_depth) || (d2 < min_depth) || (d2 > max_depth));
            if (ok && (flags & 4) && !front_off)  // LikelyFrontOfCamFilter (:124-141)
                ok = ((d1 - sqrtf(c1) * 2.f) > 0.f) && ((d2 - sqrtf(c2) * 2.f) > 0.f);
        }
        valid[n] = ok;
        local += ok;
    }
    for (int n = N + threadIdx.x; n < cap; n += blockDim.x) valid[n] = 0;
    local = wave_sum(local);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(&total, local);
    __syncthreads();
    if (threadIdx.x == 0) count[0] = total;
}

// ... addressed by lane: tables [lanes, cap, .], lane l owns rows [l cap, (l + 1) cap) (obs_filter_kernel)
__device__ __forceinline__ void obs_filter_body(const uint8_t* inbound, const double* cov1, const double* cov2, const float* vals, int flags,
                                                float min_depth, float max_depth, int cap, int N, int lane, int lanes, uint8_t* valid,
                                                int32_t* count) {
    obs_filter_rows(inbound, cov1, cov2, vals, flags, min_depth, max_depth, (size_t)lane * cap, cap, N, (size_t)lanes * cap, valid, count + lane);
}
