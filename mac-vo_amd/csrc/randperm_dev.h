// Device-side `torch.randperm(n)[:k]` — the permutation head of the keypoint selectors, drawn where the candidate count is born.
//
// Replaces the host round trip of Module/KeypointSelector.py:331,404 (`perm = torch.randperm(selected_points.size(0))[:numPoint]` on the CPU
// generator: candidate count -> D2H -> host MT19937 -> H2D / kernel arguments) with arithmetic inside the selector's finishing workgroup.  Same
// bits: torch's CPU generator is the reference MT19937 (at::mt19937) and ATen's randperm_cpu is the plain Fisher-Yates
//     r = arange(n); for i in 0 .. n - 2: z = random32() % (n - i); swap(r[i], r[i + z])          (n < 2^32 / 20)
// of which the first k outputs need the first min(k, n - 1) draws; the other draws only advance the generator.
//
// Both halves are sequential as written and parallel as restated here:
//   * MT19937 block step.  x[k + 624] = x[k + 397] ^ f(x[k], x[k + 1]) with f GF(2)-linear in the way it is used (an XOR of shifted / masked inputs), so
//     the three dependent 227-word phases of the textbook twist collapse into ONE step from the OLD block:
//         k < 227        new[k] = o[k + 397] ^ f(o[k], o[k + 1])
//         227 <= k < 454 new[k] = o[k + 170] ^ f(o[k - 227], o[k - 226]) ^ f(o[k], o[k + 1])
//         454 <= k < 623 new[k] = o[k - 57] ^ f(o[k - 454], o[k - 453]) ^ f(o[k - 227], o[k - 226]) ^ f(o[k], o[k + 1])
//         k = 623        new[623] = new[396] ^ f(o[623], new[0])      (both from the formulas above)
//     = 624 independent words, <= 7 LDS reads each, one barrier per block of 624 draws (n = 8000 candidates: 13 blocks).
//   * Partial Fisher-Yates.  With t_i = i + z_i (all known at once: the draws and the modulo are per-index work), position p is touched before step i
//     only by the steps j < i with t_j == p, and what such a step leaves there is the value w_j that position j held before step j:
//         prev(i, p) = max { j < i : t_j == p }
//         w_i   = prev(i, i) exists ? w_prev : i                    (a pointer chain, almost always empty: t_j == i needs a draw to land in the head)
//         out_i = t_i == i ? w_i : (prev(i, t_i) exists ? w_prev : t_i)
//     prev() is a lookup in an LDS hash of the <= 512 (t_j, j) pairs (chained buckets, max over the matching entries: insertion order does not matter).
// Where it runs (round 6): in the head of `backend_front_kernel` (frontend_ops.hip), i.e. on the backend stream, where consecutive frames are in order
// anyway — the generator is a sequential object and the selector segments of consecutive frames may overlap.  That kernel has ~100 workgroups and
// every one of them needs the head, so EVERY workgroup draws it (<= 2 block steps: the head's <= 512 draws span at most two blocks) and exactly ONE
// workgroup also steps through the remaining n - 1 - k draws and stores the advanced generator — into the OTHER of two state buffers, so nobody
// reads what it writes (no cross-workgroup synchronisation, no fence: DESIGN.md §5's lesson about agent-scope release / acquire beside the GEMM).
// Every function below is a PHASE: called by all threads of a workgroup (tid, nt) between barriers on the device, or by a loop over tid on the host
// (mv_randperm_heads_emulated: the CPU test suite pins this file against torch.randperm itself without a GPU).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MV_RP_FN __host__ __device__ __forceinline__
#else
#define MV_RP_FN inline
#endif

namespace mvrp {

constexpr int MT_N = 624;
constexpr int MT_STRIDE = 640;   // uint32 words per lane of the device-resident generator: [0, 624) the block, [624] the position of the next draw (624: step first)
constexpr int MAX_HEAD = 512;    // longest permutation head (num_point) the device path draws
constexpr int NBUCKET = 1024;    // hash heads (positions are spread by p & 1023: distinct small positions never share a bucket)

struct Scratch {                 // LDS on the device (13.2 KB + the heads + the caller's output row), plain memory in the host emulation
    uint32_t mt[2][MT_N];
    int32_t t[MAX_HEAD], nxt[MAX_HEAD], ptr[MAX_HEAD], hit[MAX_HEAD];
};

MV_RP_FN uint32_t mt_f(uint32_t a, uint32_t b) {
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((b & 1u) ? 0x9908b0dfu : 0u);
}

MV_RP_FN uint32_t mt_step_word(const uint32_t* o, int k) {
    if (k < 227) return o[k + 397] ^ mt_f(o[k], o[k + 1]);
    if (k < 454) return o[k + 170] ^ mt_f(o[k - 227], o[k - 226]) ^ mt_f(o[k], o[k + 1]);
    if (k < 623) return o[k - 57] ^ mt_f(o[k - 454], o[k - 453]) ^ mt_f(o[k - 227], o[k - 226]) ^ mt_f(o[k], o[k + 1]);
    const uint32_t n0 = o[397] ^ mt_f(o[0], o[1]);
    const uint32_t n396 = o[566] ^ mt_f(o[169], o[170]) ^ mt_f(o[396], o[397]);
    return n396 ^ mt_f(o[623], n0);
}

MV_RP_FN uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// init_genrand(seed) of the reference implementation (= std::mt19937(seed) = at::mt19937(seed)); position 624: the first draw steps the block
inline void mt_seed(uint32_t seed, uint32_t* state /* [MT_STRIDE] */) {
    state[0] = seed;
    for (int i = 1; i < MT_N; ++i) state[i] = 1812433253u * (state[i - 1] ^ (state[i - 1] >> 30)) + (uint32_t)i;
    state[MT_N] = MT_N;
    for (int i = MT_N + 1; i < MT_STRIDE; ++i) state[i] = 0;
}

// what one call draws: `swaps` Fisher-Yates steps give `m` outputs, the generator advances by `advance` draws
struct Plan {
    int64_t n;
    int swaps, m;
    int64_t advance;
};
MV_RP_FN Plan plan_of(int64_t n, int k) {
    Plan p;
    p.n = n;
    p.m = (int)(n < k ? (n < 0 ? 0 : n) : k);
    p.swaps = n >= 1 ? (int)(k < n - 1 ? k : n - 1) : 0;
    p.advance = n >= 1 ? n - 1 : 0;
    return p;
}

// ---- phases ----------------------------------------------------------------------------------------------------------------------------
MV_RP_FN void phase_step(const uint32_t* o, uint32_t* nw, int tid, int nt) {
    for (int k = tid; k < MT_N; k += nt) nw[k] = mt_step_word(o, k);
}

// the draws of this block that belong to the head: block words [pos, pos + take) are draws [produced, produced + take) of the call
MV_RP_FN void phase_draws(const uint32_t* x, int pos, int take, int64_t produced, const Plan& pl, int32_t* t, int tid, int nt) {
    for (int j = tid; j < take; j += nt) {
        const int64_t g = produced + j;
        if (g >= pl.swaps) break;
        t[g] = (int32_t)(g + (int64_t)(mt_temper(x[pos + j]) % (uint32_t)(pl.n - g)));
    }
}

MV_RP_FN void phase_clear(int32_t* head, int tid, int nt) {
    for (int b = tid; b < NBUCKET; b += nt) head[b] = -1;
}

#if defined(__HIP_DEVICE_COMPILE__)
#define MV_RP_XCHG(p, v) atomicExch((p), (v))
#else
static inline int32_t mv_rp_xchg_host(int32_t* p, int32_t v) { const int32_t o = *p; *p = v; return o; }
#define MV_RP_XCHG(p, v) mv_rp_xchg_host((p), (v))
#endif

MV_RP_FN void phase_insert(Scratch& s, int32_t* head, const Plan& pl, int tid, int nt) {
    for (int j = tid; j < pl.m; j += nt) {
        if (j < pl.swaps) s.nxt[j] = MV_RP_XCHG(&head[s.t[j] & (NBUCKET - 1)], j);
        else s.t[j] = j;   // (n <= k: the last output is whatever the n - 1 swaps left at position n - 1; it takes part as a step onto itself, never as a `prev`)
    }
}

MV_RP_FN int32_t prev_of(const Scratch& s, const int32_t* head, int i, int32_t p) {
    int32_t best = -1;
    for (int32_t e = head[p & (NBUCKET - 1)]; e >= 0; e = s.nxt[e])
        if (e < i && s.t[e] == p && e > best) best = e;
    return best;
}

MV_RP_FN void phase_link(Scratch& s, const int32_t* head, const Plan& pl, int tid, int nt) {
    for (int i = tid; i < pl.m; i += nt) {
        s.ptr[i] = prev_of(s, head, i, i);
        s.hit[i] = s.t[i] == i ? -2 : prev_of(s, head, i, s.t[i]);
    }
}

MV_RP_FN int32_t root_of(const Scratch& s, int32_t i) {
    while (s.ptr[i] >= 0) i = s.ptr[i];
    return i;
}

template <typename T>
MV_RP_FN void phase_emit(const Scratch& s, const Plan& pl, T* out, int tid, int nt) {
    for (int i = tid; i < pl.m; i += nt) {
        const int32_t h = s.hit[i];
        out[i] = (T)(h == -2 ? root_of(s, i) : (h >= 0 ? root_of(s, h) : s.t[i]));
    }
}

}  // namespace mvrp

#if defined(__HIPCC__)
// One call of one lane's generator by one workgroup (blockDim.x threads, all of them must call; every argument uniform).
//   state_in   the lane's MT_STRIDE words in global memory (read-only here)
//   state_out  nullptr: draw the head only.  Otherwise this workgroup also consumes the call's remaining draws and stores the advanced generator
//              there (state_out == state_in is allowed when this is the ONLY workgroup of the call: mv_randperm_head_lanes)
//   n = candidate count, k = num_point <= MAX_HEAD; writes out[0 .. min(n, k)) (LDS or global, any integer type) and returns min(n, k).
// `s` and `head` (NBUCKET words) are LDS.  Barriers: one per 624-draw block stepped + 5; the caller synchronises before it reads `out`.
template <typename T>
__device__ __forceinline__ int mv_randperm_head_wg(const uint32_t* __restrict__ state_in, uint32_t* state_out, int64_t n, int k, T* out, mvrp::Scratch& s,
                                                   int32_t* head) {
    using namespace mvrp;
    const int tid = threadIdx.x, nt = blockDim.x;
    const Plan pl = plan_of(n, k);
    phase_clear(head, tid, nt);
    if (pl.advance > 0 || state_out) {
        for (int i = tid; i < MT_N; i += nt) s.mt[0][i] = state_in[i];
        int pos = (int)state_in[MT_N];
        __syncthreads();
        int cur = 0;
        int64_t produced = 0;
        const int64_t total = state_out ? pl.advance : (int64_t)pl.swaps;   // (head-only workgroups stop behind the last draw they need)
        int64_t rem = total;
        while (rem > 0) {   // (uniform)
            if (pos == MT_N) {
                phase_step(s.mt[cur], s.mt[cur ^ 1], tid, nt);
                __syncthreads();
                cur ^= 1;
                pos = 0;
            }
            const int take = (int)(rem < MT_N - pos ? rem : MT_N - pos);
            if (produced < pl.swaps) phase_draws(s.mt[cur], pos, take, produced, pl, s.t, tid, nt);
            pos += take;
            produced += take;
            rem -= take;
        }
        if (state_out) {   // (the block's readers above and this store touch the same words read-only)
            for (int i = tid; i < MT_N; i += nt) state_out[i] = s.mt[cur][i];
            if (tid == 0) state_out[MT_N] = (uint32_t)pos;
        }
    }
    __syncthreads();   // t[] complete, heads cleared
    phase_insert(s, head, pl, tid, nt);
    __syncthreads();
    phase_link(s, head, pl, tid, nt);
    __syncthreads();
    phase_emit(s, pl, out, tid, nt);
    return pl.m;
}
#endif
