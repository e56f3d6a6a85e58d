// Device-resident keypoint permutations: the stand-alone entry points around randperm_dev.h (see there for the algorithm and what it replaces:
// Module/KeypointSelector.py:331,404 — `torch.randperm(n)[:numPoint]` on torch's CPU generator).
//   mv_mt19937_seed            host: the MT_STRIDE-word device representation of `torch.Generator().manual_seed(seed)`
//   mv_randperm_head_lanes     one workgroup per lane: n read from device memory, head + count written to device memory, generator advanced in place
//                              (the selector's finishing workgroup runs the same device function as its epilogue: mv_kp_select_draw_lanes)
//   mv_randperm_heads_emulated host-only, no GPU: the SAME phase functions executed thread by thread — pins the restatement against torch.randperm
//                              in the CPU test suite
#include "common.h"
#include "randperm_dev.h"
#include <vector>

namespace {

__global__ __launch_bounds__(1024) void randperm_head_kernel(uint32_t* __restrict__ state, const int32_t* __restrict__ n_dev, int n_stride, int k, int cap,
                                                             int64_t* __restrict__ out, int32_t* __restrict__ n_sel, int nsel_stride) {
    __shared__ mvrp::Scratch s;
    __shared__ int32_t head[mvrp::NBUCKET];
    const int l = blockIdx.x;
    uint32_t* const st = state + (size_t)l * mvrp::MT_STRIDE;
    const int m = mv_randperm_head_wg(st, st, (int64_t)n_dev[(size_t)l * n_stride], k, out + (size_t)l * cap, s, head);
    if (threadIdx.x == 0) n_sel[(size_t)l * nsel_stride] = m;
}

}  // namespace

extern "C" int mv_randperm_state_words(void) { return mvrp::MT_STRIDE; }
extern "C" int mv_randperm_max_head(void) { return mvrp::MAX_HEAD; }

extern "C" int mv_mt19937_seed(uint64_t seed, uint32_t* state_host) {
    MV_CHECK_ARG(state_host);
    mvrp::mt_seed((uint32_t)(seed & 0xffffffffull), state_host);
    return MV_OK;
}

extern "C" int mv_randperm_head_lanes(uint32_t* state, const int32_t* n_dev, int n_stride, int lanes, int k, int cap, int64_t* out_perm, int32_t* out_n_sel,
                                      int n_sel_stride, mvStream_t stream) {
    MV_CHECK_ARG(state && n_dev && out_perm && out_n_sel && lanes >= 1 && n_stride >= 1 && n_sel_stride >= 1);
    MV_CHECK_ARG(k >= 0 && k <= cap);
    if (k > mvrp::MAX_HEAD) return MV_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(randperm_head_kernel, dim3(lanes), dim3(1024), 0, (hipStream_t)stream, state, n_dev, n_stride, k, cap, out_perm, out_n_sel, n_sel_stride);
    return mv_launch_status();
}

extern "C" int mv_randperm_heads_emulated(uint64_t seed, const int64_t* n, int calls, int k, int threads, int64_t* out) {
    using namespace mvrp;
    MV_CHECK_ARG(n && out && calls >= 0 && k >= 0 && threads >= 1 && threads <= 4096);
    if (k > MAX_HEAD) return MV_ERR_UNSUPPORTED;
    std::vector<uint32_t> state(MT_STRIDE);
    mt_seed((uint32_t)(seed & 0xffffffffull), state.data());
    std::vector<int32_t> head(NBUCKET);
    Scratch* s = new Scratch;
    const int nt = threads;
#define MV_RP_ALL(call) for (int tid = 0; tid < nt; ++tid) { call; }
    for (int c = 0; c < calls; ++c) {
        MV_CHECK_ARG(n[c] >= 0 && n[c] < ((int64_t)1 << 31));
        const Plan pl = plan_of(n[c], k);
        MV_RP_ALL(phase_clear(head.data(), tid, nt));
        if (pl.advance > 0) {   // (the workgroup driver of randperm_dev.h, barriers replaced by the end of each thread loop)
            for (int i = 0; i < MT_N; ++i) s->mt[0][i] = state[i];
            int pos = (int)state[MT_N], cur = 0;
            int64_t produced = 0, rem = pl.advance;
            while (rem > 0) {
                if (pos == MT_N) {
                    MV_RP_ALL(phase_step(s->mt[cur], s->mt[cur ^ 1], tid, nt));
                    cur ^= 1;
                    pos = 0;
                }
                const int take = (int)(rem < MT_N - pos ? rem : MT_N - pos);
                if (produced < pl.swaps) MV_RP_ALL(phase_draws(s->mt[cur], pos, take, produced, pl, s->t, tid, nt));
                pos += take;
                produced += take;
                rem -= take;
            }
            for (int i = 0; i < MT_N; ++i) state[i] = s->mt[cur][i];
            state[MT_N] = (uint32_t)pos;
        }
        MV_RP_ALL(phase_insert(*s, head.data(), pl, tid, nt));
        MV_RP_ALL(phase_link(*s, head.data(), pl, tid, nt));
        MV_RP_ALL(phase_emit(*s, pl, out + (size_t)c * k, tid, nt));
    }
#undef MV_RP_ALL
    delete s;
    return MV_OK;
}
