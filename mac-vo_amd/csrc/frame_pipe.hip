// Native per-frame driver: one MACVO.run_pair of kernel launches per host call (SURVEY.md §8 A1-A22 call order)
//
// Replaces the host-side sequencing of Odometry/MACVO.py:173-311 (run_pair) + Module/Frontend/Frontend.py:215-232
// (estimate_pair) for the hot path.  The reference issues ~150 tiny torch ops per frame from Python; a Python loop over
// this library's own entry points still costs ~370 us of interpreter / ctypes time per frame, which is MORE than the
// ~330 us of GPU work (measured: the frame rate did not move when the volume GEMM got 4x faster).  Here the whole
// enqueue side is C++: ~30 launches per frame at 2-3 us each, two host calls per frame.
//
// Four HIP streams (created here, independent of the caller's) — four is the ceiling of this stack (a fifth queue costs a third of the frame rate).
// Classic layout (>= 3 lanes, mapping):
//   vol    the MFMA-bound cost-volume GEMM of frame t+1 (rotating volume buffers)
//   main   decoder side of a frame: 12 window lookups, (convex upsampling,) epilogue, dense selector, count -> host
//   back   pose-independent half of frame t's backend: gather, tracking, back-projection, covariances (one launch)
//   side   filters + rotation into the world frame + LM solve (one launch; the GPU analogue of the reference's optimizer child process,
//          Optimization/Interface.py:80-96)
// Round-5 layout (one- and two-lane pipes; `layout_alt` below): vol | main = even frames' decoder side | a second decoder-side stream = odd frames' |
//   side = backend + solve of every frame, in order; the GEMM leaves 32 CUs without a workgroup (see mv_frame_pipe_create).
// Frame t+1's frontend is enqueued before frame t's `finish`, so the selector's host round trip (candidate count ->
// torch.randperm on the CPU, kept for bit-exact indices -> permutation back) never idles the GPU.
// Stream layouts measured on the bench (ms / frame): this one 0.3235; even / odd frames each entirely on their own
// frontend stream (no event between GEMM and lookups) 0.337 — the next frame's GEMM then collides with more of the
// latency-bound kernels and both stretch; ONE frontend stream for everything 0.373 — the ~3 us launch-to-launch gaps of
// the 15 small kernels are no longer hidden under the next GEMM.
//
// Memory: every device buffer lives in ONE caller-provided arena (a torch tensor in the Python host), carved up here;
// mv_frame_pipe_buffer reports where each piece is so the host can view results without copies.  Slots rotate so a
// stage never overwrites data a still-running stage of another stream reads (maps x3, everything else x2) and the
// cross-stream hazards are closed with events (see `enqueue` / `finish`).
#include "common.h"
#include <atomic>
#include <chrono>
#include <stdio.h>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <random>
#include <thread>
#include <vector>
#include <new>
#include <stdlib.h>
#include <string.h>
#if defined(__linux__)
#include <sched.h>
#endif

#define MV_HIP(call)                                   \
    do {                                               \
        if ((call) != hipSuccess) return MV_ERR_LAUNCH; \
    } while (0)
#define MV_TRY(call)               \
    do {                           \
        const int rc_ = (call);    \
        if (rc_ != MV_OK) return rc_; \
    } while (0)

namespace {

// slot rotation for up to MAX_PENDING tracked frames in flight (enqueued, not finished): frame f's epilogue writes maps[f % N_MAPS]
// while the backend of frame f - MAX_PENDING may still read maps of f - MAX_PENDING and f - MAX_PENDING - 1
// N_CAND = MAX_PENDING + 1 (round 3): the candidate slot frame f's selector writes was last read by the backend of frame f - 4, not of the
// frame finished a moment ago — so that the enqueue of frame f never has to wait for the backend launch thread (below) to have ISSUED
// the newest finish.
#ifndef MV_MAX_PENDING
#define MV_MAX_PENDING 3
#endif
constexpr int MAX_PENDING = MV_MAX_PENDING, N_MAPS = MAX_PENDING + 2, N_CAND = MAX_PENDING + 1, N_PERM = 4, N_INEV = 8, MAX_VOL = 4, MAX_LK = 2;
constexpr int N_BEV = 8;   // ring of per-finish backend events (finish g -> slot g % N_BEV): the device-driven frame waits for the exact finish whose reads free a slot

struct Maps {
    float *disparity, *disparity_cov, *depth, *depth_cov, *match_flow, *match_cov;
    uint8_t* bad_mask;
};

struct Backend {   // every table is [lanes, cap = num_point, ...]; rows beyond a lane's n_sel are dead (valid = 0)
    int64_t *perm, *kp0;
    float *kp0f, *kp1, *vals, *sigma0, *sigma1, *pos_Tc, *pos_Tw;
    uint8_t *inbound, *valid;
    double *rot, *cov0, *cov0w, *cov1, *pose64, *info;
    int32_t* n_valid;
    int32_t* live_dev;   // [lanes, 2] device-driven frame: selected keypoints = live rows, candidate count (written by the front launch's draw)
    int32_t n_sel[MV_MAX_LANES];
};

struct Pending {
    int maps, maps_prev, cand;
    bool has_cand;
    int ti;   // timing slot of the frame (mv_frame_pipe_time_volume), -1: not timed
    long sel_job = -1;   // MV_PIPE_SELECTOR_ON=late: the finish (job) that issues this frame's selector segment; -1: already issued
    long f = -1;         // frame index (alt layout: the backend waits for the previous frame's selector segment by its event)
};

// The selector segment of a frame (upsampling / epilogue / selector(s) / count copies): everything behind the frame's last lookup.
// Issued by mv_frame_pipe_enqueue itself, or — MV_PIPE_SELECTOR_ON=vol — deferred until the NEXT frame's GEMM has been queued, so that
// it lands behind that GEMM on the GEMM's stream.
struct SelSeg {
    mvFrameInputs in;
    long f;
    int k, m, ti, maps_prev;
    bool timed, with_selector, up;
    long need_issued;   // backend launch thread: finishes that must have been issued before the segment's slot-reuse wait
};

// One `finish` split in two: the host-visible bookkeeping (slot rotation, counts, views) happens on the calling thread, the
// launches are described by this job and issued either inline or by the backend launch thread.
struct FinishJob {
    Pending pd;
    long g;                 // finish index; backend slot k = g & 1
    int pose_from, pose_to; // pose slots: prior of the frame / its optimised pose
    int n_max;
    int32_t n_sel[MV_MAX_LANES];
    int64_t n_cand[MV_MAX_LANES];   // seeded: candidate count per lane (the permutation is drawn by whoever issues the job)
    bool seeded;
    bool device = false;            // device-driven frame: the permutation is drawn inside the front launch, n_sel is an upper bound (num_point)
    std::vector<int64_t> perm;      // explicit permutations [lanes, cap] (asynchronous issue: a copy of the caller's array)
    float* pose_sink;
    double t_count = 0, t_submit = 0;   // host clock (us): candidate count seen / job queued (MV_PIPE_HOST_STATS)
    bool has_sel;                   // MV_PIPE_SELECTOR_ON=late: the selector segment of the newest enqueued frame rides behind this backend
    SelSeg sel;
};

struct Carver {   // bump allocator over the arena (or a size counter when base == nullptr)
    char* base;
    size_t off = 0;
    template <typename T>
    T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

}  // namespace

struct mvFramePipe {
    mvFramePipeConfig c;
    int plane, h8, w8, n8, KK;
    int lanes;   // independent sequences batched through every launch (= pairs / 2)
    char* arena;
    size_t arena_bytes;
    // device buffers
    float* vol[MAX_VOL];
    int n_volbuf;   // 3 (classic layout) or 4 (round-5 layout): volbufs_for
    float* tok[2 * MAX_LK];   // two alternating token buffers per decoder-side stream
    void* planes[2];   // bf16x3 split planes of fmap1 / fmap2 (volume_split3)
    // volume_split = MV_PACK_BF16X3: packed three-piece operands of the streaming split GEMM, two sets (the pack of frame f + 1 may
    // run beside the GEMM of frame f); `packed` = the shape is covered by the streaming kernel (exact fp32 kernel otherwise)
    void* pk[2][2];
    int n2t;           // cells per volume slice: n8, or mv_tiled_slice_cells(h8, w8) for a tiled fp16 volume (a padded last tile row when h8 % 4 != 0)
    void* tile16;      // vol16 && tiled: operand 2 with its pixel rows in 4 x 4-tile order (mv_fmap_tile_rows16; written and read on the GEMM's stream)
    size_t pk_bytes;
    bool packed;
    bool vol16;        // volume_split = MV_VOL_ENC16 on a shape the out16 kernel covers: 2-byte cells (Fast mode as the reference computes it)
    bool tiled;        // MV_PIPE_TILED=1 && packed: operand 2 packed in 4 x 4-tile order -> the volume is tiled for mv_corr_lookup_tiled
                       // (the driver owns both the producer and the consumer of the volume; MV_FB_VOLUME then shows that layout)
    int pack_on;       // 0 = on the GEMM's stream (in front of it), 1 = backend stream, 2 = decoder-side stream
    int sel_on_back;   // 1: upsampling / epilogue / selector / count copy of a frame run on the backend stream behind its lookups
    hipEvent_t e_lk[N_CAND];   // a frame's last lookup (decoder-side stream)
    hipEvent_t e_packed[2];
    float *up_flow, *up_cov;
    Maps maps[N_MAPS];
    void* kp_ws;
    size_t kp_ws_bytes;
    int32_t* cand[N_CAND];
    int32_t* count[N_CAND];
    float* stats[N_CAND];
    Backend be[2];
    float* pose[3];   // [lanes, 7]
    float *intr, *bl;   // [lanes, 4], [lanes]
    int32_t* offs;    // [lanes + 1]: lane l owns rows [l * cap, (l + 1) * cap) of the backend tables
    int fuse_backend; // 1 (default): backend = mv_backend_front_lanes + mv_pgo_solve_posed, two launches; MV_PIPE_FUSE_BACKEND=0: the five-launch form
    // host
    int32_t* h_count[N_CAND];      // pinned
    int64_t* h_perm[N_PERM];  // pinned
    // streams / events
    hipStream_t s_vol, s_main, s_back, s_side;
    hipStream_t s_lk[MAX_LK];   // decoder-side streams: frame f's lookups run on s_lk[f % n_lk]; s_lk[0] = s_main
    int n_lk;
    int alt;                     // the two-decoder-stream layout: see mv_frame_pipe_create
    int free_cus;                // CUs the volume GEMM leaves without a persistent workgroup (mv_corr_volume_packed_shared)
    hipEvent_t e_seg[N_INEV];    // alt: end of frame f's selector segment, slot f % N_INEV (the next frame's segment runs on the other decoder-side stream)
    bool seg_valid[N_INEV];
    int alt_indep;               // alt: consecutive frames' segments may overlap (own selector workspace each; NODEPTH selector, no upsampling)
    void* kp_ws2;                // ... the odd frames' workspace
    int lean_chain;              // device-driven frames with the front launch on the decoder side: fewer packets on the frame's chain — no e_cand / e_seg markers behind
                                 // the selector (nobody waits for them: the front launch follows in order), the volume buffer is released by the front launch's event
                                 // instead of a marker behind the lookups.  A marker costs the queue ~6 us, a cross-queue barrier ~10 (rocprofv3 trace, r06_chain_gaps.log)
    bool cand_recorded[N_CAND];  // e_cand[k] holds this frame's selector segment (lean chain: recorded only when somebody asks)
    hipEvent_t vol_free_ev[MAX_VOL];   // what the GEMM that rewrites volume buffer k waits for (e_vol_free[k], or the ring event of the frame's front launch)
    int front_on_decoder;        // device-driven alt layout: a frame's front launch (draw + gathers + covariances) runs on the frame's decoder-side stream right
                                 // behind its selector segment (no cross-queue barrier in front of it); the backend stream carries the solves only

    hipStream_t s_sel;   // MV_PIPE_SELECTOR_ON=own: a fifth stream for the selector segment (nullptr otherwise)
    SelSeg deferred;     // MV_PIPE_SELECTOR_ON=vol: the newest frame's selector segment, not issued yet
    bool deferred_valid;
    hipEvent_t e_rest[N_INEV];   // inputs of the decoder side (coords, flow, ...) when the GEMM was issued ahead of them
    hipEvent_t e_in[N_INEV], e_vol_done[MAX_VOL], e_vol_free[MAX_VOL], e_cand[N_CAND], e_backend[N_BEV], e_pgo, e_perm[N_PERM];
    hipEvent_t e_release;     // the consumer's reads of result views enqueued so far (mv_frame_pipe_release)
    bool release_valid;
    hipEvent_t e_posed[2];    // backend slot k: world-frame tables written (side stream, in front of the solve)
    hipEvent_t e_solved[2];   // backend slot k: its solve has finished reading the tables
    bool vol_free_valid[MAX_VOL], backend_valid[N_BEV], pgo_valid, perm_valid[N_PERM], solved_valid[2];
    // state
    long n_enq, n_fin;
    long n_vol;            // volume GEMMs issued (n_enq <= n_vol <= n_enq + 1: at most one GEMM ahead of its frame's decoder side)
    hipEvent_t e_in_of[MAX_VOL];   // per volume buffer: the input-ready event its GEMM waited for (the decoder side re-uses it)
    int lookups_on_main;   // default 1; MV_PIPE_LOOKUPS_ON=vol is the measured alternative
    int pose_cur;
    int prior_slot;        // pose slot the newest finished frame started from (its motion-model prior)
    hipEvent_t e_map;
    int newest_maps;
    std::deque<Pending> pending;
    // dense-mapping tail (config.mapping; lanes == 1)
    int32_t* cand_m[N_CAND];
    int32_t* count_m[N_CAND];
    float* stats_m[N_CAND];
    int32_t* h_count_m[N_CAND];    // pinned
    int32_t* h_nvalid[2];          // pinned: observation count of backend slot k
    hipEvent_t e_nvalid[2];
    bool nvalid_valid[2];
    int64_t *mp_perm, *mp_uv;      // [map_num_point], [map_num_point, 2]
    float *mp_uvf, *mp_d, *mp_sdd, *mp_sigma, *mp_Tc, *mp_Tw;
    double* mp_cov;
    uint8_t* mp_color;
    int64_t* h_perm_m;             // pinned
    hipEvent_t e_maptail;          // the map tail's last kernel (buffers + pinned permutation reusable)
    bool maptail_valid;
    int mp_rows;                   // rows of the newest map tail (0: none for the newest finished frame)
    int last_maps_prev, last_cand;   // of the newest finished frame
    // native keypoint permutations (mv_frame_pipe_seed_lanes): one MT19937 per lane, the engine behind torch's CPU generator
    std::vector<std::mt19937> rng;
    std::vector<int32_t> perm_scratch;   // identity array of the partial Fisher-Yates, reused
    std::vector<int64_t> perm_host;      // [lanes, cap]
    std::vector<int32_t> nsel_host;
    // Device-driven frame (round 6, mv_frame_pipe_seed_lanes with MV_PIPE_DEVICE_DRAW != 0): the same generators live in DEVICE memory and the permutation head is
    // drawn inside the backend's front launch (randperm_dev.h) from the count the selector left in device memory — no D2H count, no host wait, no host draw, no
    // H2D permutation: enqueue + finish of a frame are a fixed chain of launches.  Two state buffers: finish g reads rp_state[g & 1], writes rp_state[(g + 1) & 1].
    uint32_t* rp_state[2];   // [lanes, mv_randperm_state_words()]
    int dev_draw;
    long last_backend_frame = -1;   // frame index of the newest backend issued with keypoints (launch-thread / inline issuer only)
    // optional timing of the dominant kernel (bench.py roofline): event pairs around each volume GEMM on its stream
    int vol_timed[MAX_VOL];      // timing slot of the GEMM that filled each volume buffer (-1: not timed)
    std::vector<hipEvent_t> tv0, tv1, tv2, tv3;   // GEMM start / end, last lookup done, selector done (timeline hook)
    std::vector<hipEvent_t> tv4, tv5, tv6, tv7;   // backend start / end (backend stream), pose_apply start / solve end (solve stream)
    int n_timed, timed_cap;
    int time_detail;   // 1 (default): a timed frame also records the six timeline events tv2..tv7 on three streams; 0: only the pair around its GEMM
    // Backend launch thread (round 3).  A one-lane stream is bound by the HOST: ~38 launches per frame at ~4 us each on one thread
    // (tools/host_breakdown.py: 172 us of host time per frame, 6 us of it waiting for the GPU).  With `async_backend` the ~10
    // launches of `finish` + the permutation draw are issued by this thread while the caller's thread already enqueues the next
    // frame's decoder side.  Everything that is not enqueue / enqueue_volume / wait_candidates / finish / release / buffer first
    // drains the job queue (`flush_jobs`), so the two threads never touch the same piece of driver state:
    //   caller's thread: pending, n_enq / n_fin / pose_cur / views, s_vol + s_main launches, the selector segment on s_back
    //   launch thread:   Backend tables, e_solved / e_posed / e_pgo / e_perm / e_backend, pinned permutation slots, rng
    // Cross-thread event edges: e_cand and e_release are recorded by the caller BEFORE the job is queued; e_backend is consumed by
    // the caller's enqueue only after `issued` says the launch thread has recorded it.
    int async_backend;
    int async_explicit;     // the caller / MV_PIPE_ASYNC_BACKEND chose (the device-driven pipe otherwise drops the launch thread: see mv_frame_pipe_seed_lanes)
    int device;
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::deque<FinishJob> jobs;
    long issued;            // finishes whose launches have all been issued (guarded by mu)
    std::atomic<long> n_submitted{0};   // jobs ever queued: what the launch thread spins on
    std::atomic<bool> stop_flag{false};
    double spin_us;         // how long the launch thread spins for the next job before it sleeps (MV_PIPE_LAUNCH_SPIN_US, default 250)
    int host_stats;         // MV_PIPE_HOST_STATS=1: mean host latencies of the selector-count -> backend-launch chain, printed by destroy
    double st_n = 0, st_count_to_submit = 0, st_submit_to_pick = 0, st_pick_to_issued = 0, st_wait = 0;
    double st_enq = 0, st_enq_lookups = 0, st_enq_seg = 0, st_vol = 0, st_fin = 0, st_calls = 0;   // us inside enqueue (its lookups / its selector segment), enqueue_volume, finish_device
    bool stop;
    int async_rc;           // first error of an asynchronously issued job, reported by the next call
};

// cross-stream dependency; when the event has already fired no barrier packet is queued at all (every
// hipStreamWaitEvent costs the waiting queue a few microseconds even for a long-fired event)
static int wait_if_pending(hipStream_t s, hipEvent_t e) {
    const hipError_t q = hipEventQuery(e);
    if (q == hipSuccess) return MV_OK;
    if (q != hipErrorNotReady) return MV_ERR_LAUNCH;
    return hipStreamWaitEvent(s, e, 0) == hipSuccess ? MV_OK : MV_ERR_LAUNCH;
}

static int affinity_cores() {
#if defined(__linux__)
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) return CPU_COUNT(&set);
#endif
    const unsigned n = std::thread::hardware_concurrency();
    return n ? (int)n : 1;
}

static inline double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

#define MV_ASYNC_DEFAULT(p) (1)   // measured (640x480, f16x2 volume): one lane 6.01 k vs 5.26 k frames/s, 32 lanes 7.56 k vs 7.54 k
static int wait_issued(mvFramePipe* p, long n);
static int flush_deferred(mvFramePipe* p);
static int issue_selector_segment(mvFramePipe* p, const SelSeg& d);
static int flush_jobs(mvFramePipe* p);
static void launch_thread_main(mvFramePipe* p);

// 1 = the round-5 layout for one- and two-lane pipes (two decoder-side streams; see mv_frame_pipe_create).  MV_PIPE_LAYOUT=classic | alt; an explicit
// placement knob of the classic layout (MV_PIPE_SELECTOR_ON, MV_PIPE_LOOKUPS_ON=vol) selects the classic layout
static int layout_alt(int lanes, int mapping) {
    const char* e = getenv("MV_PIPE_LAYOUT");
    if (e && strcmp(e, "classic") == 0) return 0;
    const char* lk = getenv("MV_PIPE_LOOKUPS_ON");
    if (lanes > 2 || mapping || (lk && strcmp(lk, "vol") == 0) || getenv("MV_PIPE_SELECTOR_ON")) return 0;
    return 1;
}

// tracked frames the host should keep in flight (pipeline.NativeHotPath.run): the alt layout and batched pipes want 3, the classic one-lane pipe 2
extern "C" int mv_frame_pipe_default_depth(int lanes, int mapping) { return (lanes > 2 || layout_alt(lanes, mapping)) ? 3 : 2; }

static int volbufs_for(int lanes, int mapping) {
    // 3: with a GEMM issued one frame ahead (mv_frame_pipe_enqueue_volume) the buffer it rewrites was last read by the lookups of frame t - 1, long
    // finished; with 2 (rounds 1-2; an A/B knob until round 5) it waited for frame t's lookups, which run beside the previous GEMM at a third of their
    // isolated speed (measured 272 vs 245 us per frame)
    // alt layout: three tracked frames in flight + the GEMM one ahead = 4 (measured 6.95 k vs 6.74 k frames/s at 300 steps with 3, the same at 20; 5: no gain)
    return layout_alt(lanes, mapping) ? 4 : 3;
}

static size_t carve(mvFramePipe* p, char* base) {
    const mvFramePipeConfig& c = p->c;
    Carver a{base};
    const size_t L = p->lanes;
    const size_t plane = p->plane, n8 = p->n8, B = c.pairs, N = c.num_point > 0 ? c.num_point : 1;
    for (int k = 0; k < p->n_volbuf; ++k) p->vol[k] = a.take<float>(B * n8 * n8);
    for (int k = 0; k < 2 * (L <= 2 ? MAX_LK : 1); ++k) p->tok[k] = a.take<float>(B * p->KK * n8);
    for (int k = 0; k < 2; ++k)
        p->planes[k] = (c.volume_split == 2 || c.volume_split == 3) ? (void*)a.take<uint16_t>(3 * B * n8 * c.C) : nullptr;
    p->pk_bytes = p->packed ? mv_volume_pack_bytes((int)B, c.C, (int)n8, c.volume_split) : 0;
    for (int k = 0; k < 2; ++k)
        for (int o = 0; o < 2; ++o) p->pk[k][o] = p->packed ? (void*)a.take<char>(p->pk_bytes) : nullptr;
    // (carved for every Fast-mode pipe whose shape the tiled form covers — the sizing call knows the configuration, not MV_PIPE_TILED: 2 % of the volume buffers)
    p->tile16 = (c.volume_split == MV_VOL_ENC16 && c.radius == 4 && p->w8 % 4 == 0) ? (void*)a.take<uint16_t>(B * (size_t)mv_tiled_slice_cells(p->h8, p->w8) * c.C) : nullptr;
    p->up_flow = a.take<float>(B * 2 * plane);
    p->up_cov = a.take<float>(B * 2 * plane);
    for (int k = 0; k < N_MAPS; ++k) {   // every map is [lanes, ch, H, W]
        Maps& m = p->maps[k];
        m.disparity = a.take<float>(L * plane);
        m.disparity_cov = a.take<float>(L * plane);
        m.depth = a.take<float>(L * plane);
        m.depth_cov = a.take<float>(L * plane);
        m.match_flow = a.take<float>(L * 2 * plane);
        m.match_cov = a.take<float>(L * 3 * plane);
        m.bad_mask = a.take<uint8_t>(L * plane);
    }
    p->kp_ws_bytes = L * mv_kp_select_workspace_bytes(c.H, c.W);
    p->kp_ws = a.take<char>(p->kp_ws_bytes);
    p->kp_ws2 = L <= 2 ? a.take<char>(p->kp_ws_bytes) : nullptr;

    for (int k = 0; k < N_CAND; ++k) {
        p->cand[k] = a.take<int32_t>(L * plane);
        p->count[k] = a.take<int32_t>(L * 4);
        p->stats[k] = a.take<float>(L * 4);
    }
    for (int k = 0; k < 2; ++k) {
        Backend& b = p->be[k];
        b.perm = a.take<int64_t>(L * N);
        b.kp0 = a.take<int64_t>(L * 2 * N);
        b.kp0f = a.take<float>(L * 2 * N);
        b.kp1 = a.take<float>(L * 2 * N);
        b.vals = a.take<float>(L * 11 * N);
        b.sigma0 = a.take<float>(L * 3 * N);
        b.sigma1 = a.take<float>(L * 3 * N);
        b.pos_Tc = a.take<float>(L * 3 * N);
        b.pos_Tw = a.take<float>(L * 3 * N);
        b.inbound = a.take<uint8_t>(L * N);
        b.valid = a.take<uint8_t>(L * N);
        b.rot = a.take<double>(L * 9);
        b.cov0 = a.take<double>(L * 9 * N);
        b.cov0w = a.take<double>(L * 9 * N);
        b.cov1 = a.take<double>(L * 9 * N);
        b.pose64 = a.take<double>(L * 7);
        b.info = a.take<double>(L * 4);
        b.n_valid = a.take<int32_t>(L);
        b.live_dev = a.take<int32_t>(2 * L);
    }
    for (int k = 0; k < 2; ++k) p->rp_state[k] = a.take<uint32_t>(L * (size_t)mv_randperm_state_words());
    if (c.mapping) {
        const size_t MP = c.map_num_point > 0 ? c.map_num_point : 1;
        for (int k = 0; k < N_CAND; ++k) {
            p->cand_m[k] = a.take<int32_t>(plane);
            p->count_m[k] = a.take<int32_t>(4);
            p->stats_m[k] = a.take<float>(4);
        }
        p->mp_perm = a.take<int64_t>(MP);
        p->mp_uv = a.take<int64_t>(2 * MP);
        p->mp_uvf = a.take<float>(2 * MP);
        p->mp_d = a.take<float>(MP);
        p->mp_sdd = a.take<float>(MP);
        p->mp_sigma = a.take<float>(3 * MP);
        p->mp_Tc = a.take<float>(3 * MP);
        p->mp_Tw = a.take<float>(3 * MP);
        p->mp_cov = a.take<double>(9 * MP);
        p->mp_color = a.take<uint8_t>(3 * MP);
    }
    for (int k = 0; k < 3; ++k) p->pose[k] = a.take<float>(L * 7);
    p->intr = a.take<float>(L * 4);
    p->bl = a.take<float>(L);
    p->offs = a.take<int32_t>(L + 1);
    return (a.off + 255) & ~(size_t)255;
}

static int check_config(const mvFramePipeConfig* c) {
    MV_CHECK_ARG(c);
    MV_CHECK_ARG(c->H > 0 && c->W > 0 && c->H % 8 == 0 && c->W % 8 == 0);
    MV_CHECK_ARG(c->C > 0 && c->C % 16 == 0 && c->iters >= 0);
    MV_CHECK_ARG(c->pairs >= 2 && c->pairs % 2 == 0 && c->pairs / 2 <= MV_MAX_LANES);   // lane l = pairs 2l (stereo), 2l + 1 (temporal)
    MV_CHECK_ARG(c->radius >= 1 && c->radius <= 4);
    MV_CHECK_ARG(c->selector_mode == MV_KP_NODEPTH || c->selector_mode == MV_KP_FULL);
    MV_CHECK_ARG(c->num_point >= 0 && c->edgewidth >= 0 && c->min_num_point >= 0);
    MV_CHECK_ARG(c->graph_type >= MV_GRAPH_ICP && c->graph_type <= MV_GRAPH_DISP);
    MV_CHECK_ARG(c->mapping == 0 || (c->mapping == 1 && c->pairs == 2 && c->map_num_point > 0 && c->map_mask_width >= 0));
    MV_CHECK_ARG(c->volume_split == 0 || c->volume_split == 2 || c->volume_split == 3 || c->volume_split == MV_PACK_BF16X3 ||
                 c->volume_split == MV_PACK_F16X2 || c->volume_split == MV_VOL_ENC16);
    MV_CHECK_ARG(!c->volume_split || c->volume_split == MV_VOL_ENC16 || c->feat_dtype == MV_F32);
    MV_CHECK_ARG(c->volume_split != MV_VOL_ENC16 || (c->feat_dtype == MV_F16 && c->radius == 4));   // (the lookup reads fp16 cells)
    MV_CHECK_ARG(!(c->volume_split == 2 || c->volume_split == 3) || c->layout == MV_LAYOUT_HWC);   // (the packed form takes either layout)
    return MV_OK;
}

extern "C" size_t mv_frame_pipe_arena_bytes(const mvFramePipeConfig* cfg) {
    if (check_config(cfg) != MV_OK) return 0;
    mvFramePipe tmp{};
    tmp.c = *cfg;
    tmp.plane = cfg->H * cfg->W;
    tmp.h8 = cfg->H / 8;
    tmp.w8 = cfg->W / 8;
    tmp.n8 = tmp.h8 * tmp.w8;
    tmp.KK = (2 * cfg->radius + 1) * (2 * cfg->radius + 1);
    tmp.lanes = cfg->pairs / 2;
    tmp.n_volbuf = volbufs_for(tmp.lanes, cfg->mapping);
    tmp.packed = (cfg->volume_split == MV_PACK_BF16X3 || cfg->volume_split == MV_PACK_F16X2) &&
                 mv_corr_volume_packed_supported(cfg->pairs, cfg->C, tmp.n8, tmp.n8, cfg->volume_split);
    return carve(&tmp, nullptr);
}

extern "C" int mv_frame_pipe_max_pending(void) { return MAX_PENDING; }

extern "C" void mv_frame_pipe_destroy(mvFramePipe* p) {
    if (!p) return;
    if (p->worker.joinable()) {
        {
            std::lock_guard<std::mutex> lk(p->mu);
            p->stop = true;
        }
        p->stop_flag.store(true, std::memory_order_relaxed);
        p->cv_job.notify_all();
        p->worker.join();   // (drains the queue first)
    }
    if (p->host_stats && p->st_calls > 0)
        fprintf(stderr, "[mv_frame_pipe host stats] per frame over %.0f device-driven frames: enqueue %.1f us (lookups %.1f, selector segment %.1f), enqueue_volume %.1f us, "
                        "finish_device %.1f us\n", p->st_calls, p->st_enq / p->st_calls, p->st_enq_lookups / p->st_calls, p->st_enq_seg / p->st_calls,
                p->st_vol / p->st_calls, p->st_fin / p->st_calls);
    if (p->host_stats && p->st_n > 0)
        fprintf(stderr, "[mv_frame_pipe host stats] finishes %.0f: wait for the candidate count %.1f us, count seen -> job queued %.1f us, queued -> picked up by the "
                        "launch thread %.1f us, picked up -> every launch issued %.1f us (means per frame)\n",
                p->st_n, p->st_wait / p->st_n, p->st_count_to_submit / p->st_n, p->st_submit_to_pick / p->st_n, p->st_pick_to_issued / p->st_n);
    (void)hipStreamSynchronize(p->s_vol);
    for (int k = 0; k < MAX_LK; ++k) if (p->s_lk[k]) (void)hipStreamSynchronize(p->s_lk[k]);
    (void)hipStreamSynchronize(p->s_back);
    (void)hipStreamSynchronize(p->s_side);
    if (p->s_sel) (void)hipStreamSynchronize(p->s_sel);
    auto ev = [](hipEvent_t e) { if (e) (void)hipEventDestroy(e); };
    for (auto e : p->e_in) ev(e);
    for (auto e : p->e_rest) ev(e);
    for (int k = 0; k < MAX_VOL; ++k) { ev(p->e_vol_done[k]); ev(p->e_vol_free[k]); }
    for (int k = 0; k < N_CAND; ++k) ev(p->e_cand[k]);
    for (int k = 0; k < 2; ++k) { ev(p->e_posed[k]); ev(p->e_solved[k]); }
    for (auto e : p->e_backend) ev(e);
    ev(p->e_pgo);
    for (int k = 0; k < N_CAND; ++k) ev(p->e_lk[k]);
    for (auto e : p->e_seg) ev(e);
    ev(p->e_maptail);
    for (int k = 0; k < 2; ++k) { ev(p->e_nvalid[k]); if (p->h_nvalid[k]) (void)hipHostFree(p->h_nvalid[k]); }
    for (int k = 0; k < N_CAND; ++k) if (p->h_count_m[k]) (void)hipHostFree(p->h_count_m[k]);
    if (p->h_perm_m) (void)hipHostFree(p->h_perm_m);
    ev(p->e_packed[0]);
    ev(p->e_packed[1]);
    ev(p->e_map);
    ev(p->e_release);
    for (auto e : p->e_perm) ev(e);
    for (auto e : p->tv0) ev(e);
    for (auto e : p->tv1) ev(e);
    for (auto e : p->tv2) ev(e);
    for (auto e : p->tv3) ev(e);
    for (auto e : p->tv4) ev(e);
    for (auto e : p->tv5) ev(e);
    for (auto e : p->tv6) ev(e);
    for (auto e : p->tv7) ev(e);
    for (int k = 0; k < N_CAND; ++k) if (p->h_count[k]) (void)hipHostFree(p->h_count[k]);
    for (auto h : p->h_perm) if (h) (void)hipHostFree(h);
    if (p->s_vol) (void)hipStreamDestroy(p->s_vol);
    for (int k = 0; k < MAX_LK; ++k) if (p->s_lk[k]) (void)hipStreamDestroy(p->s_lk[k]);   // (s_lk[0] = s_main)
    if (p->s_back && p->s_back != p->s_side) (void)hipStreamDestroy(p->s_back);
    if (p->s_side) (void)hipStreamDestroy(p->s_side);
    if (p->s_sel) (void)hipStreamDestroy(p->s_sel);
    delete p;
}

static int create_impl(mvFramePipe* p) {
    const mvFramePipeConfig& c = p->c;
    int lo = 0, hi = 0;
    MV_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));   // hi = numerically lowest = highest priority
    // the GEMM and the decoder side run at normal priority; the short pose-dependent kernels go first when slots free up
    // (rounds 3-4 A/B, removed in round 5: confining the GEMM / the small kernels to disjoint CU sets with hipExtStreamCreateWithCUMask — any mask cost the
    // GEMM 12 % and the small kernels, grids sized for the whole chip, far more: 0.60-0.83 vs 0.32 ms per frame — and a raised queue priority for the
    // decoder-side stream: no gain.  profiles/r03_*; DESIGN.md changelog.)
    MV_HIP(hipStreamCreateWithPriority(&p->s_vol, hipStreamNonBlocking, 0));
    MV_HIP(hipStreamCreateWithPriority(&p->s_main, hipStreamNonBlocking, 0));
    p->s_lk[0] = p->s_main;
    for (int k = 1; k < p->n_lk; ++k) MV_HIP(hipStreamCreateWithPriority(&p->s_lk[k], hipStreamNonBlocking, 0));
    // (creation order matters on this stack: with `side` created before `back` the same schedule ran 15 % slower, profiles/r05_pipe_ab.log)
    if (!p->alt) MV_HIP(hipStreamCreateWithPriority(&p->s_back, hipStreamNonBlocking, hi));
    MV_HIP(hipStreamCreateWithPriority(&p->s_side, hipStreamNonBlocking, hi));
    if (p->alt) p->s_back = p->s_side;   // backend + solve of a frame in order on ONE stream: the fourth queue belongs to the odd frames' decoder side
    if (p->sel_on_back == 2) MV_HIP(hipStreamCreateWithPriority(&p->s_sel, hipStreamNonBlocking, hi));
    auto mk = [](hipEvent_t* e) { return hipEventCreateWithFlags(e, hipEventDisableTiming); };
    for (auto& e : p->e_in) MV_HIP(mk(&e));
    for (auto& e : p->e_rest) MV_HIP(mk(&e));
    for (int k = 0; k < MAX_VOL; ++k) {
        MV_HIP(mk(&p->e_vol_done[k]));
        MV_HIP(mk(&p->e_vol_free[k]));
    }
    for (int k = 0; k < N_CAND; ++k) {
        MV_HIP(mk(&p->e_cand[k]));
        MV_HIP(hipHostMalloc((void**)&p->h_count[k], (size_t)p->lanes * 4 * sizeof(int32_t), hipHostMallocDefault));
    }
    for (auto& e : p->e_backend) MV_HIP(mk(&e));
    for (int k = 0; k < 2; ++k) {
        MV_HIP(mk(&p->e_posed[k]));
        MV_HIP(mk(&p->e_solved[k]));
    }
    MV_HIP(mk(&p->e_pgo));
    for (int k = 0; k < N_CAND; ++k) MV_HIP(mk(&p->e_lk[k]));
    for (auto& e : p->e_seg) MV_HIP(mk(&e));
    MV_HIP(mk(&p->e_maptail));
    for (int k = 0; k < 2; ++k) {
        MV_HIP(mk(&p->e_nvalid[k]));
        MV_HIP(hipHostMalloc((void**)&p->h_nvalid[k], (size_t)p->lanes * sizeof(int32_t), hipHostMallocDefault));
    }
    if (c.mapping) {
        for (int k = 0; k < N_CAND; ++k) MV_HIP(hipHostMalloc((void**)&p->h_count_m[k], 4 * sizeof(int32_t), hipHostMallocDefault));
        MV_HIP(hipHostMalloc((void**)&p->h_perm_m, (size_t)c.map_num_point * sizeof(int64_t), hipHostMallocDefault));
    }
    MV_HIP(mk(&p->e_packed[0]));
    MV_HIP(mk(&p->e_packed[1]));
    MV_HIP(mk(&p->e_map));
    MV_HIP(mk(&p->e_release));
    const size_t N = c.num_point > 0 ? c.num_point : 1;
    for (int k = 0; k < N_PERM; ++k) {
        MV_HIP(mk(&p->e_perm[k]));
        MV_HIP(hipHostMalloc((void**)&p->h_perm[k], (size_t)p->lanes * N * sizeof(int64_t), hipHostMallocDefault));
    }
    // constants: selector workspace zeroed once (mv_kp_select leaves it zeroed), identity pose, PGO scalars, offsets table
    MV_HIP(hipMemsetAsync(p->kp_ws, 0, p->kp_ws_bytes, p->s_main));
    if (p->kp_ws2) MV_HIP(hipMemsetAsync(p->kp_ws2, 0, p->kp_ws_bytes, p->s_main));
    const int L = p->lanes;
    std::vector<float> ident((size_t)L * 7, 0.f), intr((size_t)L * 4), bl((size_t)L, c.baseline);
    std::vector<int32_t> offs((size_t)L + 1);
    for (int l = 0; l < L; ++l) {
        ident[7 * l + 6] = 1.f;
        intr[4 * l] = c.fx; intr[4 * l + 1] = c.fy; intr[4 * l + 2] = c.cx; intr[4 * l + 3] = c.cy;
    }
    for (int l = 0; l <= L; ++l) offs[l] = (int32_t)(l * N);
    MV_HIP(hipMemcpyAsync(p->pose[0], ident.data(), ident.size() * sizeof(float), hipMemcpyHostToDevice, p->s_main));
    MV_HIP(hipMemcpyAsync(p->intr, intr.data(), intr.size() * sizeof(float), hipMemcpyHostToDevice, p->s_main));
    MV_HIP(hipMemcpyAsync(p->bl, bl.data(), bl.size() * sizeof(float), hipMemcpyHostToDevice, p->s_main));
    MV_HIP(hipMemcpyAsync(p->offs, offs.data(), offs.size() * sizeof(int32_t), hipMemcpyHostToDevice, p->s_main));
    { const char* e = getenv("MV_PIPE_FUSE_BACKEND"); p->fuse_backend = (e && atoi(e) == 0) ? 0 : 1; }
    p->time_detail = 1;
    MV_HIP(hipStreamSynchronize(p->s_main));   // the host vectors die here
    return MV_OK;
}

extern "C" int mv_frame_pipe_create(const mvFramePipeConfig* cfg, void* arena, size_t arena_bytes, mvFramePipe** out) {
    MV_CHECK_ARG(out && arena);
    MV_TRY(check_config(cfg));
    mvFramePipe* p = new (std::nothrow) mvFramePipe{};
    if (!p) return MV_ERR_WORKSPACE;
    p->c = *cfg;
    p->plane = cfg->H * cfg->W;
    p->h8 = cfg->H / 8;
    p->w8 = cfg->W / 8;
    p->n8 = p->h8 * p->w8;
    p->KK = (2 * cfg->radius + 1) * (2 * cfg->radius + 1);
    p->lanes = cfg->pairs / 2;
    p->n_volbuf = volbufs_for(p->lanes, cfg->mapping);
    p->packed = (cfg->volume_split == MV_PACK_BF16X3 || cfg->volume_split == MV_PACK_F16X2) &&
                mv_corr_volume_packed_supported(cfg->pairs, cfg->C, p->n8, p->n8, cfg->volume_split);
    // shapes outside the out16 kernel's domain keep the fp32-stored volume of the same 16-bit GEMM
    p->vol16 = cfg->volume_split == MV_VOL_ENC16 && mv_corr_volume_out16_supported(cfg->pairs, cfg->C, p->n8, p->n8, cfg->feat_dtype, cfg->layout);
    {
        // Tiled volume for the batched lookups, MV_PIPE_TILED=1.  Measured (640x480, 32 lanes, f16x2): the tiled lookup
        // alone is 15 % faster (B = 64: 114 -> 97 us: the same bytes in half as many, aligned 64-byte requests), the 32-lane step
        // 6 % SLOWER (7.44 k vs 7.95 k frames/s; beside the GEMM's write stream the row-major form's 32-byte sector gathers do better)
        // -> off by default.
        // [r6] ... and ON by default for Fast-mode pipes: measured 32 lanes 11.8 k -> 13.3 k frames/s, 8 lanes 12.0 k -> 13.3 k, one lane 8.3 k -> 8.4-8.6 k
        const char* e = getenv("MV_PIPE_TILED");
        const bool want = e ? atoi(e) != 0 : p->vol16;
        // [r6] Fast mode's 2-byte cells: a tile is one 32-byte sector (mv_corr_lookup_tiled_vol16: B = 64 alone 115 -> 79 us)
        p->tiled = want && cfg->radius == 4 && (p->w8 % 4) == 0 &&
                   (p->vol16 ? mv_corr_volume_out16_supported(cfg->pairs, cfg->C, p->n8, mv_tiled_slice_cells(p->h8, p->w8), cfg->feat_dtype, cfg->layout) != 0
                             : (p->packed && (p->h8 % 4) == 0));
        p->n2t = (p->tiled && p->vol16) ? mv_tiled_slice_cells(p->h8, p->w8) : p->n8;    // (2-byte cells: a padded slice still fits the buffer's n8 * n8 * 4 bytes)
    }
    {
        // MV_PIPE_SELECTOR_ON=back: the selector segment of a frame (epilogue, NMS, finishing workgroup, count copy: ~60 us beside the
        // GEMM) moves from the decoder-side stream — which a one-lane stream saturates: 12 dependent lookups + that segment = one
        // period — to the backend stream, behind an event on the frame's last lookup; the next frame's lookups start meanwhile.
        // Measured (640x480, f16x2 volume): one lane 5.38 k vs 4.94 k frames/s (period 183 vs 198 us), 32 lanes 7.43 k vs 7.64 k: the
        // default follows the lane count; MV_PIPE_SELECTOR_ON=main|back forces it.
        // [r4] `late` needs two unfinished frames in front of the new one, i.e. a 3-deep pipeline; the default depth for <= 2 lanes is now 2
        // (pipeline.py; profiles/r04_latency_ab.log: the same frame rate within the box's +-3 %, GEMM start -> pose 1.18 -> 0.82 ms), where
        // `late` and `back` are the same schedule.
        const char* e = getenv("MV_PIPE_SELECTOR_ON");
        p->sel_on_back = e ? (strcmp(e, "back") == 0 ? 1 : strcmp(e, "own") == 0 ? 2 : strcmp(e, "vol") == 0 ? 3 : strcmp(e, "late") == 0 ? 4 : 0)
                           : (p->lanes <= 2 ? 1 : 0);
    }
    {
        // Where the operand pack of frame f + 1 runs.  It needs only the feature maps, so it can run beside GEMM(f) on another of
        // the pipe's streams (a fifth stream measured 2.50 k vs 3.41 k frames/s in round 2: four is the ceiling on this stack).
        const char* e = getenv("MV_PIPE_PACK_ON");
        // Measured (640x480, one lane, frames/s): in front of the GEMM on its own stream 4.28 k, backend stream 4.09 k, decoder-side
        // stream 3.11 k: beside a one-wave-per-SIMD GEMM every co-running kernel costs the GEMM more than the 8 us the pack takes.
        p->pack_on = (e && strcmp(e, "back") == 0) ? 1 : (e && strcmp(e, "main") == 0) ? 2 : (e && strcmp(e, "side") == 0) ? 3 : 0;
    }
    p->arena = (char*)arena;
    p->arena_bytes = arena_bytes;
    if (((uintptr_t)arena & 255) != 0 || carve(p, p->arena) > arena_bytes) {
        delete p;
        return MV_ERR_WORKSPACE;
    }
    p->newest_maps = -1;
    for (auto& r : p->cand_recorded) r = true;
    {
        const char* e = getenv("MV_PIPE_LOOKUPS_ON");
        p->lookups_on_main = (e && strcmp(e, "vol") == 0) ? 0 : 1;
    }
    {
        // Decoder-side streams (round 5).  The 12 dependent lookups of a frame are latency-bound (5.8 us each alone, 13-15 us per launch beside the GEMM): on
        // ONE stream they are 160-175 us per frame, and that stream — busy all the time — was the period of a one-lane pipe.  Consecutive frames' decoder
        // sides are independent (own volume buffer, coordinates, token buffers, maps slot, candidate slot), so frame f's lookups run on stream f % n_lk.
        //   * A FIFTH queue is not an option on this stack: with one more stream every configuration ran at 3.4-3.8 k frames/s instead of 5.9 k (GEMM 89 ->
        //     132 us; also with GPU_MAX_HW_QUEUES=8; profiles/r05_pipe_ab.log run 15, and the same finding in round 2; the A/B knob is gone).
        //   * The alt layout keeps FOUR:   vol: pack + GEMM | main: even frames' lookups + selector segment | a second decoder-side stream: the odd
        //     frames' | side: backend + solve of every frame, in order (the chain solve -> solve is sequential anyway; the backend no longer overlaps it).
        //     Consecutive selector segments may overlap (own selector workspace per parity); the backend waits for the previous frame's segment by event.
        p->alt = layout_alt(p->lanes, cfg->mapping);
        p->n_lk = p->alt ? 2 : 1;
        if (p->alt) p->sel_on_back = 0;
        // ... and in that layout the GEMM leaves 32 CUs (4 per XCD) without a persistent workgroup: backend + solve share ONE stream there, and the solve's
        // workgroup (496 registers + 77 KB of LDS) cannot sit beside a GEMM wave (300 registers, the LDS ring) — with no CU free it waited for the gap between two GEMMs (73 of its 127 us).  Measured
        // (profiles/r05_pipe_ab.log): 0 / 8 / 16 / 32 / 48 / 64 free = 6.26 / 6.34 / 6.68 / 6.84 / 6.87 / 6.82 k frames/s; in the classic layout no gain (r3, r5).
        p->free_cus = p->alt ? 32 : 0;
        if (const char* e = getenv("MV_PIPE_FREE_CUS")) p->free_cus = atoi(e);   // A/B knob (multiples of 8: per XCD)
        p->alt_indep = p->alt;   // (ordered segments for every selector: -6 % on the 20-step line, profiles/r05_pipe_ab.log run 18)
        // Device-driven frames: the front launch (permutation draw + gathers + both covariance models) rides behind the frame's own selector segment on its
        // decoder-side stream — no cross-queue barrier in front of it — and the fourth stream carries the solves only.  With front + solve in order on one
        // stream that stream was the bound of a device-driven pipe (a frame's backend started 250 us after its selector had finished: a backlog of front /
        // solve pairs, each with a pending barrier in front).  Measured (profiles/r06_device_draw_ab.log): 20 steps 5.97 / 5.63 / 6.18 k vs 5.75 / 5.65 / 5.61 k
        // frames/s, 300 steps 6.83-6.95 k vs 6.60-6.64 k; bit-identical.  MV_PIPE_FRONT_ON=side: the old placement.  (Also measured and dropped: the selector's
        // FINISHING workgroup moved in front of the backend on the backend's stream — 5.62 k at 300 steps.)
        // (Also measured and dropped, same log: THREE decoder-side streams — the fourth stream as the third, a frame's whole chain incl. its solve in order on one
        // stream, solves chained across streams by event: bit-identical, 5.3 k instead of 6.7 k frames/s at 300 steps.  More concurrency is not what this pipe lacks.)
        { const char* e = getenv("MV_PIPE_LEAN"); p->lean_chain = (p->alt && !(e && atoi(e) == 0)) ? 1 : 0; }
        { const char* e = getenv("MV_PIPE_FRONT_ON"); p->front_on_decoder = (p->alt && !(e && strcmp(e, "side") == 0)) ? 1 : 0; }
    }
    const int rc = create_impl(p);
    if (rc != MV_OK) {
        mv_frame_pipe_destroy(p);
        return rc;
    }
    {
        // config.async_backend: 1 on, -1 off, 0 = MV_PIPE_ASYNC_BACKEND if set, else the measured default (see mvFramePipe)
        const char* e = getenv("MV_PIPE_ASYNC_BACKEND");
        const int want = cfg->async_backend ? cfg->async_backend : (e ? (atoi(e) ? 1 : -1) : MV_ASYNC_DEFAULT(p));
        p->async_backend = want > 0 ? 1 : 0;
        p->async_explicit = (cfg->async_backend || e) ? 1 : 0;
        if (hipGetDevice(&p->device) != hipSuccess) p->async_backend = 0;
        {
            // a spinning launch thread needs a core of its own beside the caller's (and Python's): with fewer than three in the process' affinity mask it
            // sleeps on the condition variable instead (per-rank core slices of bench --gpus N, 1-2 core hosts)
            const char* e3 = getenv("MV_PIPE_LAUNCH_SPIN_US");
            p->spin_us = e3 ? atof(e3) : (affinity_cores() >= 3 ? 250.0 : 0.0);
        }
        { const char* e4 = getenv("MV_PIPE_HOST_STATS"); p->host_stats = e4 ? atoi(e4) : 0; }
        if (p->async_backend) p->worker = std::thread(launch_thread_main, p);
    }
    *out = p;
    return MV_OK;
}

extern "C" int mv_frame_pipe_set_pose(mvFramePipe* p, const float* pose7_host) {
    MV_CHECK_ARG(p && pose7_host);
    MV_TRY(flush_jobs(p));
    if (p->pgo_valid) MV_HIP(hipEventSynchronize(p->e_pgo));
    MV_HIP(hipMemcpy(p->pose[p->pose_cur], pose7_host, (size_t)p->lanes * 7 * sizeof(float), hipMemcpyHostToDevice));
    return MV_OK;
}

// ------------------------------------------------------------------------------------------------ frontend half
typedef int (*mvLookupFn)(const float*, const float*, float*, int, int, int, int, int, int, mvStream_t);
static int lookup_vol16_as_float_ptr(const float* vol, const float* coords, float* out, int B, int H1, int W1, int H2, int W2, int radius, mvStream_t s) {
    return mv_corr_lookup_vol16(vol, coords, out, B, H1, W1, H2, W2, radius, s);      // (the arena pointer is typed float*; the cells are fp16)
}
static int lookup_tiled_vol16_as_float_ptr(const float* vol, const float* coords, float* out, int B, int H1, int W1, int H2, int W2, int radius, mvStream_t s) {
    return mv_corr_lookup_tiled_vol16(vol, coords, out, B, H1, W1, H2, W2, radius, s);
}
static mvLookupFn lookup_of(const mvFramePipe* p) {
    if (p->vol16) return p->tiled ? lookup_tiled_vol16_as_float_ptr : lookup_vol16_as_float_ptr;
    return p->tiled ? mv_corr_lookup_tiled : mv_corr_lookup;
}

// volume GEMM of frame n_vol on its own stream; a buffer is rewritten only after the lookups that read it have finished
static int issue_volume(mvFramePipe* p, const mvFrameInputs* in, mvStream_t in_stream) {
    const mvFramePipeConfig& c = p->c;
    const long f = p->n_vol;
    const int k = (int)(f % p->n_volbuf);
    const int B = c.pairs;
    // inputs were produced on the caller's stream
    hipEvent_t e_in = p->e_in[f % N_INEV];
    MV_HIP(hipEventRecord(e_in, (hipStream_t)in_stream));
    p->e_in_of[k] = e_in;
    MV_TRY(wait_if_pending(p->s_vol, e_in));
    if (p->vol_free_valid[k]) MV_TRY(wait_if_pending(p->s_vol, p->vol_free_ev[k] ? p->vol_free_ev[k] : p->e_vol_free[k]));
    const bool timed = p->n_timed < p->timed_cap;
    if (timed && !p->packed) MV_HIP(hipEventRecord(p->tv0[p->n_timed], p->s_vol));
    if (p->packed) {
        void** pk = p->pk[f & 1];
        hipStream_t sp = p->pack_on == 0 ? p->s_vol : p->pack_on == 1 ? p->s_back : p->pack_on == 2 ? p->s_main : p->s_side;
        if (sp != p->s_vol) {
            MV_TRY(wait_if_pending(sp, e_in));
            // this operand set was last read by the GEMM of frame f - 2 (same stream order as the volume buffers' events)
            if (f >= 2) MV_TRY(wait_if_pending(sp, p->e_vol_done[(f - 2) % p->n_volbuf]));
        }
        if (p->tiled)
            MV_TRY(mv_volume_pack_tiled((const float*)in->fmap1, (const float*)in->fmap2, pk[0], pk[1], B, c.C, p->n8, p->h8, p->w8,
                                        c.layout, c.volume_split, sp));
        else
            MV_TRY(mv_volume_pack((const float*)in->fmap1, (const float*)in->fmap2, pk[0], pk[1], B, c.C, p->n8, p->n8, c.layout,
                                  c.volume_split, sp));
        if (sp != p->s_vol) {
            MV_HIP(hipEventRecord(p->e_packed[f & 1], sp));
            MV_HIP(hipStreamWaitEvent(p->s_vol, p->e_packed[f & 1], 0));
        }
        // (round-4 A/B, removed: the GEMM waiting for the previous frame's lookups — each then has the chip to itself — cost 21 % of the frame rate)
        if (timed) MV_HIP(hipEventRecord(p->tv0[p->n_timed], p->s_vol));   // the GEMM alone: behind the pack, wherever that ran
        MV_TRY(mv_corr_volume_packed_shared(pk[0], pk[1], p->vol[k], B, c.C, p->n8, p->n8, c.volume_split, p->free_cus, p->s_vol));
    } else if (c.volume_split == 2 || c.volume_split == 3) {
        const size_t nel = (size_t)B * p->n8 * c.C;
        MV_TRY(mv_split_bf16x3((const float*)in->fmap1, p->planes[0], nel, p->s_vol));
        MV_TRY(mv_split_bf16x3((const float*)in->fmap2, p->planes[1], nel, p->s_vol));
        MV_TRY(mv_corr_volume(p->planes[0], p->planes[1], p->vol[k], B, c.C, p->n8, p->n8,
                              c.volume_split == 2 ? MV_BF16X2 : MV_BF16X3, MV_LAYOUT_HWC, p->s_vol));
    } else if (p->vol16) {
        const void* op2 = in->fmap2;
        if (p->tiled) {
            MV_TRY(mv_fmap_tile_rows16(in->fmap2, p->tile16, B, c.C, p->h8, p->w8, p->s_vol));
            op2 = p->tile16;
        }
        MV_TRY(mv_corr_volume_out16(in->fmap1, op2, p->vol[k], B, c.C, p->n8, p->n2t, c.feat_dtype, c.layout, p->s_vol));
    } else {
        MV_TRY(mv_corr_volume(in->fmap1, in->fmap2, p->vol[k], B, c.C, p->n8, p->n8, c.feat_dtype, c.layout, p->s_vol));
    }
    p->vol_timed[k] = timed ? p->n_timed : -1;
    if (timed) MV_HIP(hipEventRecord(p->tv1[p->n_timed++], p->s_vol));
    MV_HIP(hipEventRecord(p->e_vol_done[k], p->s_vol));
    p->n_vol = f + 1;
    return MV_OK;
}

// Optional early issue of the NEXT frame's volume GEMM (frame n_enq, before its mv_frame_pipe_enqueue): the GEMM only
// needs the feature maps and a free volume buffer, so the host can queue it one frame ahead — before it blocks on the
// previous frame's candidate count — and the GEMM stream never waits for the host (measured: ~93 us of idle per frame).
extern "C" int mv_frame_pipe_enqueue_volume(mvFramePipe* p, const mvFrameInputs* in, mvStream_t in_stream) {
    MV_CHECK_ARG(p && in && in->fmap1 && in->fmap2);
    MV_CHECK_ARG(p->n_vol == p->n_enq);                 // at most one GEMM ahead of its frame
    MV_CHECK_ARG(p->lookups_on_main);                   // the alternative layout keeps the lookups behind the GEMM on its stream
    // the volume buffer of frame n_vol was last read by the lookups of frame n_vol - 2: their event exists (frame enqueued)
    const double t_hs = p->host_stats ? now_us() : 0.0;
    MV_TRY(issue_volume(p, in, in_stream));
    if (p->host_stats) p->st_vol += now_us() - t_hs;
    // MV_PIPE_SELECTOR_ON=vol: the newest enqueued frame's selector segment goes behind this GEMM on the GEMM's stream — it needs that
    // frame's lookups, which finish about when this GEMM does, and then runs in the gap the GEMM stream idles in anyway, alone on the chip
    return p->sel_on_back == 3 ? flush_deferred(p) : MV_OK;
}

// lean chain: the markers behind frame f's selector segment were not recorded; whoever needs them (a host-permuted finish, mv_frame_pipe_wait_candidates, a
// selector segment that depends on the previous one) records them now, on the frame's decoder-side stream — a later point of the same in-order stream
static int ensure_chain_events(mvFramePipe* p, long f, int cand_slot) {
    if (cand_slot >= 0 && !p->cand_recorded[cand_slot]) {
        MV_HIP(hipEventRecord(p->e_cand[cand_slot], p->s_lk[f % p->n_lk]));
        p->cand_recorded[cand_slot] = true;
    }
    if (p->alt && f >= 0 && !p->seg_valid[f % N_INEV]) {
        MV_HIP(hipEventRecord(p->e_seg[f % N_INEV], p->s_lk[f % p->n_lk]));
        p->seg_valid[f % N_INEV] = true;
    }
    return MV_OK;
}

static int issue_selector_segment(mvFramePipe* p, const SelSeg& d) {
    const mvFramePipeConfig& c = p->c;
    const mvFrameInputs* in = &d.in;
    const int k = d.k, m = d.m, ti = d.ti, B = c.pairs;
    const bool timed = d.timed, with_selector = d.with_selector, up = d.up;
    hipStream_t s = p->s_lk[d.f % p->n_lk];   // (same stream as the frame's lookups: in order behind them)
    if (p->sel_on_back) {   // everything behind the lookups continues on another stream (in order with the backends it must follow)
        s = p->sel_on_back == 2 ? p->s_sel : p->sel_on_back == 3 ? p->s_vol : p->s_back;   // (1 and 4: the backend stream)
        MV_HIP(hipStreamWaitEvent(s, p->e_lk[k], 0));
    }
    // maps slot m and candidate slot k were last read by the backend of frame f - 2 (f - 3 for the maps) on `back`
    // (the newest backend event covers the older one: same stream)
    if (p->async_backend) {
        // Maps slot m and candidate slot k were last read by the backend of frame f - 4.  Of the three tracked frames behind it,
        // pending.size() are not finished yet and the others are the newest finishes: frame f - 4 is finish number
        // n_fin - 3 + pending.size() - 1, which the launch thread issued long ago (this wait does not block in steady state); any
        // backend event recorded at or after it orders this stream behind it (same stream).
        if (!with_selector) MV_TRY(flush_jobs(p));   // (re-)initialisation: no assumption about what is in flight
        long issued;
        {
            const long need = d.need_issued;
            MV_TRY(wait_issued(p, need < 0 ? 0 : need));
            std::lock_guard<std::mutex> lk(p->mu);
            issued = p->issued;
        }
        // (device-driven frame: a finish is issued right behind its frame's enqueue, so "the newest issued backend" would be the PREVIOUS frame's and the
        // segments would run one after the other; the ring holds the exact finish — frame f - 4's, long done — instead)
        const long w = p->dev_draw ? d.need_issued - 1 : issued - 1;
        if (w >= 0 && w < issued) MV_TRY(wait_if_pending(s, p->e_backend[w % N_BEV]));
    } else if (p->n_fin > 0) {
        const long w = p->dev_draw ? d.need_issued - 1 : p->n_fin - 1;
        if (w >= 0 && p->backend_valid[w % N_BEV]) MV_TRY(wait_if_pending(s, p->e_backend[w % N_BEV]));
    }
    if (p->release_valid) MV_TRY(wait_if_pending(s, p->e_release));   // ... and by consumers of result views (mv_frame_pipe_release)
    // alt: the previous frame's segment ran on the OTHER decoder-side stream; this one reads its maps (FULL selector, and the backend's gathers
    // behind e_cand) and shares the selector workspace / upsampling buffers with it
    // (alt_indep: NODEPTH selector without upsampling reads nothing of the previous frame and has its own workspace — the segments may overlap and
    // the BACKEND waits for the previous frame's segment instead, finish_issue)
    const bool indep = p->alt_indep && !up && c.selector_mode == MV_KP_NODEPTH;
    if (p->alt && !indep && d.f > 0) {
        MV_TRY(ensure_chain_events(p, d.f - 1, -1));
        MV_TRY(wait_if_pending(s, p->e_seg[(d.f - 1) % N_INEV]));
    }
    void* const kp_ws = (indep && (d.f & 1)) ? p->kp_ws2 : p->kp_ws;
    Maps& mp = p->maps[m];
    constexpr bool fuse_epi = true;   // epilogue + the selector's first kernel in one launch (the separate form was an A/B knob of rounds 2-4)
    if (up) {
        MV_TRY(mv_convex_upsample(in->flow8, in->up_mask, p->up_flow, B, p->h8, p->w8, 0.25f, 0, s));
        MV_TRY(mv_convex_upsample(in->cov8, in->cov_mask, p->up_cov, B, p->h8, p->w8, 1.0f, 1, s));
        MV_TRY(mv_frontend_epilogue_lanes(p->up_flow, p->up_cov, 0, c.H, c.W, c.bl_fx, c.bl_fx_sq, mp.disparity,
                                          mp.disparity_cov, mp.depth, mp.depth_cov, nullptr, mp.match_flow, mp.match_cov,
                                          p->lanes, s));
    } else if (!(fuse_epi && with_selector && c.selector_mode == MV_KP_NODEPTH)) {
        MV_TRY(mv_frontend_epilogue_lanes(in->flow, in->logcov, 1, c.H, c.W, c.bl_fx, c.bl_fx_sq, mp.disparity,
                                          mp.disparity_cov, mp.depth, mp.depth_cov, nullptr, mp.match_flow, mp.match_cov,
                                          p->lanes, s));
    }
    const Pending pd{m, d.maps_prev, k, with_selector, ti};
    if (with_selector) {
        mvKpSelectParams sp{c.H, c.W, c.selector_mode, c.kp_kernel_size, c.kp_mask_width, c.max_depth, c.max_depth_cov,
                            c.max_match_cov};
        if (c.selector_mode == MV_KP_NODEPTH && fuse_epi && !up) {
            // epilogue + selector's first kernel in one launch (one launch and one pass over the maps less on the chain that
            // bounds a single-sequence stream)
            MV_TRY(mv_frontend_epilogue_select_lanes(in->flow, in->logcov, 1, c.bl_fx, c.bl_fx_sq, mp.disparity, mp.disparity_cov,
                                                     mp.depth, mp.depth_cov, nullptr, mp.match_flow, mp.match_cov, nullptr,
                                                     nullptr, &sp, kp_ws, p->kp_ws_bytes, p->cand[k], p->count[k],
                                                     p->stats[k], p->lanes, s));
        } else if (c.selector_mode == MV_KP_NODEPTH) {
            MV_TRY(mv_kp_select_lanes(mp.match_cov, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &sp, kp_ws,
                                      p->kp_ws_bytes, p->cand[k], p->count[k], p->stats[k], p->lanes, s));
        } else {
            const Maps& m0 = p->maps[pd.maps_prev];
            MV_TRY(mv_kp_select_lanes(mp.match_cov, m0.depth, m0.depth_cov, mp.depth, mp.depth_cov, nullptr, nullptr, &sp,
                                      p->kp_ws, p->kp_ws_bytes, p->cand[k], p->count[k], p->stats[k], p->lanes, s));
        }
        if (!p->dev_draw)   // (the device-driven frame reads the count where it is)
            MV_HIP(hipMemcpyAsync(p->h_count[k], p->count[k], (size_t)p->lanes * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        if (c.mapping) {
            // MappingPointSelector works on the PREVIOUS frame's depth maps (KeypointSelector.py:87-100; MACVO.py:315): its count
            // travels to the host with the tracking selector's
            const Maps& m0 = p->maps[pd.maps_prev];
            mvKpSelectParams spm{c.H, c.W, MV_KP_MAPPING, c.kp_kernel_size, c.map_mask_width, c.map_max_depth, c.map_max_depth_cov,
                                 c.max_match_cov};
            MV_TRY(mv_kp_select_lanes(nullptr, m0.depth, m0.depth_cov, nullptr, nullptr, nullptr, nullptr, &spm, p->kp_ws,
                                      p->kp_ws_bytes, p->cand_m[k], p->count_m[k], p->stats_m[k], 1, s));
            MV_HIP(hipMemcpyAsync(p->h_count_m[k], p->count_m[k], 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        }
        // lean chain (device-driven, front launch in order behind this segment, segments independent): no marker here — e_cand is recorded on demand
        // (ensure_cand_event) and nothing waits for e_seg
        const bool lean = p->lean_chain && p->dev_draw && p->front_on_decoder && indep && !p->sel_on_back && !c.mapping;
        p->cand_recorded[k] = !lean;
        if (!lean) MV_HIP(hipEventRecord(p->e_cand[k], s));
        if (timed) MV_HIP(hipEventRecord(p->tv3[ti], s));
        if (p->alt) {
            if (!lean) MV_HIP(hipEventRecord(p->e_seg[d.f % N_INEV], s));
            p->seg_valid[d.f % N_INEV] = !lean;
        }
    } else if (p->alt) {
        MV_HIP(hipEventRecord(p->e_seg[d.f % N_INEV], s));
        p->seg_valid[d.f % N_INEV] = true;
    }
    return MV_OK;
}

static int flush_deferred(mvFramePipe* p) {
    if (!p->deferred_valid) return MV_OK;
    p->deferred_valid = false;
    return issue_selector_segment(p, p->deferred);
}
// in front of a wait for / finish of the OLDEST pending frame: its selector segment must have been issued.  MV_PIPE_SELECTOR_ON=late keeps
// the newest frame's segment for the finish that is about to happen, unless that newest frame IS the oldest pending one; a segment that
// travelled with an earlier finish has been issued once the launch thread is through with that job.
static int selector_of_front_issued(mvFramePipe* p) {
    if (p->sel_on_back != 4 || p->pending.size() <= 1) MV_TRY(flush_deferred(p));
    const long sj = p->pending.empty() ? -1 : p->pending.front().sel_job;
    return sj >= 0 ? wait_issued(p, sj + 1) : MV_OK;
}

extern "C" int mv_frame_pipe_enqueue(mvFramePipe* p, const mvFrameInputs* in, mvStream_t in_stream, int with_selector) {
    MV_CHECK_ARG(p && in && in->fmap1 && in->fmap2);
    const mvFramePipeConfig& c = p->c;
    MV_CHECK_ARG(c.iters == 0 || in->coords);
    const bool up = in->flow8 != nullptr;
    MV_CHECK_ARG(up ? (in->cov8 && in->up_mask && in->cov_mask) : (in->flow && in->logcov));
    const long f = p->n_enq;
    const int k = (int)(f % N_CAND), m = (int)(f % N_MAPS);
    MV_CHECK_ARG(!with_selector || p->newest_maps >= 0);   // a tracked frame needs the previous frame's maps
    MV_CHECK_ARG(!with_selector || (int)p->pending.size() < MAX_PENDING);  // slot rotation covers MAX_PENDING tracked frames in flight
    const int B = c.pairs;
    const double t_hs0 = p->host_stats ? now_us() : 0.0;
    MV_TRY(flush_deferred(p));   // (the previous frame's selector segment, had nobody asked for it yet: frames stay in order)

    // ---- volume GEMM (own stream) unless mv_frame_pipe_enqueue_volume already issued it
    const bool ahead = p->n_vol != f;
    if (!ahead) MV_TRY(issue_volume(p, in, in_stream));
    const int kv = (int)(f % p->n_volbuf);   // volume buffer of this frame (k = its candidate / backend slot)
    const int ti = p->time_detail ? p->vol_timed[kv] : -1;   // timeline slot (the GEMM's own event pair was recorded by issue_volume)
    const bool timed = ti >= 0;

    // ---- decoder side on `main`, overlapping the next frame's GEMM.  (MV_PIPE_LOOKUPS_ON=vol keeps the lookups on the
    // GEMM's stream instead — no cross-stream event in front of the first lookup, GEMM undisturbed at 215 us — measured
    // 0.3256 vs 0.3197 ms per frame: behind a 215-us kernel each of the 12 launch boundaries costs ~9 us.)
    const size_t coord_stride = (size_t)B * 2 * p->n8;
    const int lk = (int)(f % p->n_lk);
    hipStream_t s = p->s_lk[lk];
    float* const* tok = p->tok + 2 * lk;
    if (ahead) {   // the GEMM's input event predates this call: order the decoder side after the caller's stream as of NOW
        hipEvent_t e = p->e_rest[f % N_INEV];
        MV_HIP(hipEventRecord(e, (hipStream_t)in_stream));
        MV_TRY(wait_if_pending(s, e));
    }
    if (p->lookups_on_main) {
        MV_HIP(hipStreamWaitEvent(s, p->e_vol_done[kv], 0));   // also orders `s` after e_in (the GEMM stream waited for it)
        for (int it = 0; it < c.iters; ++it)
            MV_TRY(lookup_of(p)(p->vol[kv], in->coords + it * coord_stride, tok[it & 1], B, p->h8, p->w8, p->h8, p->w8, c.radius, s));
        // lean chain: this frame's front launch (same stream, behind the lookups) releases the buffer — its ring event exists before the GEMM that rewrites the
        // buffer is issued (that GEMM belongs to frame f + n_volbuf, enqueued only after this frame has been finished: MAX_PENDING < n_volbuf)
        const bool lean = p->lean_chain && p->dev_draw && with_selector && c.num_point > 0 && p->n_volbuf > MAX_PENDING;
        if (lean) {
            p->vol_free_ev[kv] = p->e_backend[(p->n_fin + (long)p->pending.size()) % N_BEV];
        } else {
            MV_HIP(hipEventRecord(p->e_vol_free[kv], s));
            p->vol_free_ev[kv] = p->e_vol_free[kv];
        }
        p->vol_free_valid[kv] = true;
        if (timed) MV_HIP(hipEventRecord(p->tv2[ti], s));
    } else {
        for (int it = 0; it < c.iters; ++it)
            MV_TRY(lookup_of(p)(p->vol[kv], in->coords + it * coord_stride, tok[it & 1], B, p->h8, p->w8, p->h8, p->w8, c.radius, p->s_vol));
        MV_HIP(hipEventRecord(p->e_vol_done[kv], p->s_vol));   // volume AND its lookups done
        MV_HIP(hipStreamWaitEvent(s, p->e_vol_done[kv], 0));   // also orders `s` after e_in
        p->vol_free_valid[kv] = false;                         // vol[k] / tok are only touched on s_vol: stream order suffices
    }

    const double t_hs1 = p->host_stats ? now_us() : 0.0;
    SelSeg d{*in, f, k, m, ti, p->newest_maps, timed, with_selector != 0, up,
             p->n_fin - MAX_PENDING + (long)p->pending.size()};
    if (p->sel_on_back) MV_HIP(hipEventRecord(p->e_lk[k], s));   // the segment continues on another stream behind the last lookup
    // MV_PIPE_SELECTOR_ON=late: with two unfinished frames in front of it (the steady state of a 3-deep pipeline) the segment is
    // handed to the NEXT finish, which issues it right behind that frame's backend: the backend stream then holds
    // backend(f - 2), selector(f), backend(f - 1), selector(f + 1) ... — a backend waits behind ONE queued selector segment (whose
    // lookups are done by then) instead of two that still wait for their lookups (frame latency: see DESIGN, backend timeline)
    const bool late = p->sel_on_back == 4 && with_selector && p->pending.size() >= 2;
    if ((p->sel_on_back == 3 && with_selector) || late) {
        p->deferred = d;            // issued behind the next frame's GEMM (mv_frame_pipe_enqueue_volume) / the next backend, or by whoever needs it first
        p->deferred_valid = true;
    } else {
        MV_TRY(issue_selector_segment(p, d));
    }
    if (with_selector) p->pending.push_back(Pending{m, p->newest_maps, k, true, ti, -1, f});
    p->newest_maps = m;
    p->n_enq = f + 1;
    if (p->host_stats) {
        const double t_hs2 = now_us();
        p->st_enq += t_hs2 - t_hs0;
        p->st_enq_lookups += t_hs1 - t_hs0;
        p->st_enq_seg += t_hs2 - t_hs1;
    }
    return MV_OK;
}

extern "C" int mv_frame_pipe_wait_candidates(mvFramePipe* p, int32_t* n_cand) {
    MV_CHECK_ARG(p && n_cand && !p->pending.empty());
    MV_TRY(selector_of_front_issued(p));
    const Pending& pd = p->pending.front();
    MV_TRY(ensure_chain_events(p, pd.f, pd.cand));
    MV_HIP(hipEventSynchronize(p->e_cand[pd.cand]));
    if (p->dev_draw)   // (the device-driven pipe has no per-frame count copy)
        MV_HIP(hipMemcpy(p->h_count[pd.cand], p->count[pd.cand], (size_t)p->lanes * 4 * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int l = 0; l < p->lanes; ++l) n_cand[l] = p->h_count[pd.cand][4 * l];
    return MV_OK;
}

// ------------------------------------------------------------------------------------------------ native permutations
// `selected[torch.randperm(n)[:numPoint]]` (Module/KeypointSelector.py:331,404) draws from torch's CPU generator, which is a
// standard MT19937 (at::mt19937, init_genrand(seed & 0xffffffff)), and torch.randperm on the CPU is the plain Fisher-Yates
//     r = arange(n); for i in 0 .. n - 2: z = random32() % (n - i); swap(r[i], r[i + z])          (ATen randperm_cpu, n < 2^32 / 20)
// Position i is final after iteration i, so the first numPoint outputs need numPoint swaps; the other n - 1 - numPoint draws only
// advance the generator.  One lane of the reference's own call costs ~45 us at n = 8000; 32 lanes are 1.4-3 ms per step on the
// host — more than the 2.6 ms volume GEMM of the whole batch leaves idle (measured: 1.9 ms of GEMM-stream idle per 32-lane step).
// Here: ~2 ns per draw, no allocation.  Identical bits to `torch.Generator().manual_seed(seed)` + torch.randperm
// (tests/test_gpu_lanes.py::test_native_seeded_lanes_equal_torch_generators).
extern "C" int mv_frame_pipe_seed_lanes(mvFramePipe* p, const uint64_t* seeds) {
    MV_CHECK_ARG(p && seeds);
    MV_TRY(flush_jobs(p));   // the launch thread's draw_perms reads rng / perm_host
    p->rng.clear();
    for (int l = 0; l < p->lanes; ++l) p->rng.emplace_back((uint32_t)(seeds[l] & 0xffffffffull));
    const int cap = p->c.num_point > 0 ? p->c.num_point : 1;
    p->perm_host.assign((size_t)p->lanes * cap, 0);
    p->nsel_host.assign((size_t)p->lanes, 0);
    {
        // ... and the same generators in device memory: the device-driven frame (default wherever it applies; MV_PIPE_DEVICE_DRAW=0: the host draw above).
        // Needs the two-launch backend, a head the device draw covers, and no dense-mapping tail (its second permutation is drawn by the Python side).
        const char* e = getenv("MV_PIPE_DEVICE_DRAW");
        const bool want = e ? atoi(e) != 0 : true;
        p->dev_draw = want && p->fuse_backend && !p->c.mapping && p->c.num_point >= 1 && p->c.num_point <= mv_randperm_max_head() && p->pending.empty();
        if (p->dev_draw && p->async_backend && !p->async_explicit && affinity_cores() < 3) {
            // One host thread on small hosts.  The launch thread existed to overlap the host draw + the backend launches with the caller's next enqueue while the caller
            // waited for candidate counts; a device-driven frame has no wait and no draw, and one thread issues it in ~105-140 us — about the frame period.  With the lean
            // chain (below) that is the edge: measured (profiles/r06_chain_gaps.log) one thread 6.99 / 6.85 / 6.64 k frames/s at 300 steps with issue times of 115 / 134 /
            // 143 us, with the launch thread taking the two backend launches (caller 92-103 us) 6.90 / 6.97 / 6.93 k and no low outliers on the 20-step line; pinned to one
            // or two cores both forms run at 5.9-6.1 k (the host-drawn frame: 3.9 k on one core).  So: the launch thread stays where the process has >= 3 cores.
            MV_TRY(flush_jobs(p));
            {
                std::lock_guard<std::mutex> lk(p->mu);
                p->stop = true;
            }
            p->stop_flag.store(true, std::memory_order_relaxed);
            p->cv_job.notify_all();
            if (p->worker.joinable()) p->worker.join();
            p->async_backend = 0;
        }
        if (p->dev_draw) {
            const size_t W = (size_t)mv_randperm_state_words();
            std::vector<uint32_t> st((size_t)p->lanes * W);
            for (int l = 0; l < p->lanes; ++l) MV_TRY(mv_mt19937_seed(seeds[l], st.data() + (size_t)l * W));
            MV_HIP(hipStreamSynchronize(p->s_back));   // (no front launch may be reading a generator: they run on the backend stream or on the
            for (int k = 0; k < p->n_lk; ++k) MV_HIP(hipStreamSynchronize(p->s_lk[k]));   //  decoder-side streams)
            MV_HIP(hipMemcpy(p->rp_state[p->n_fin & 1], st.data(), st.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
    }
    return MV_OK;
}

extern "C" int mv_frame_pipe_device_draw(const mvFramePipe* p) { return p && p->dev_draw ? 1 : 0; }
extern "C" int mv_frame_pipe_volume_tiled(const mvFramePipe* p) { return p && p->tiled ? 1 : 0; }   // the volume buffers' slices are in 4 x 4-cell tiles
extern "C" int mv_frame_pipe_host_threads(const mvFramePipe* p) { return p ? (p->async_backend ? 2 : 1) : 0; }   // the caller's (+ the backend launch thread)

static void randperm_head(std::mt19937& eng, int64_t n, int k, std::vector<int32_t>& r, int64_t* out) {
    if (n <= 0) return;
    if ((int64_t)r.size() < n) r.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) r[(size_t)i] = (int32_t)i;
    const int64_t swaps = k < n - 1 ? k : n - 1;
    for (int64_t i = 0; i < swaps; ++i) {
        const int64_t z = (int64_t)((uint32_t)eng()) % (n - i);
        const int32_t sav = r[(size_t)i];
        r[(size_t)i] = r[(size_t)(i + z)];
        r[(size_t)(i + z)] = sav;
    }
    if (n - 1 > swaps) eng.discard((unsigned long long)(n - 1 - swaps));   // the draws of the remaining iterations
    const int64_t m = k < n ? k : n;
    for (int64_t i = 0; i < m; ++i) out[i] = r[(size_t)i];
}

// ------------------------------------------------------------------------------------------------ pose-dependent half
// Host part of a finish: validate, take the frame off the pending list, rotate the slots and update everything the caller's
// thread reads afterwards (views, counts).  No HIP call.
static int finish_host(mvFramePipe* p, const int32_t* n_sel, float* pose_sink, FinishJob& j) {
    const mvFramePipeConfig& c = p->c;
    const int L = p->lanes;
    int n_max = 0;
    for (int l = 0; l < L; ++l) {
        MV_CHECK_ARG(n_sel[l] >= 0 && n_sel[l] <= c.num_point);
        n_max = n_sel[l] > n_max ? n_sel[l] : n_max;
    }
    j.pd = p->pending.front();
    p->pending.pop_front();
    j.g = p->n_fin;
    j.has_sel = false;
    if (p->sel_on_back == 4 && p->deferred_valid && !p->pending.empty()) {   // (never the frame being finished: flush_deferred ran first)
        j.sel = p->deferred;
        j.has_sel = true;
        p->deferred_valid = false;
        p->pending.back().sel_job = j.g;
    }
    j.n_max = n_max;
    j.pose_sink = pose_sink;
    Backend& b = p->be[j.g & 1];
    for (int l = 0; l < L; ++l) b.n_sel[l] = j.n_sel[l] = n_sel[l];
    p->n_fin = j.g + 1;
    p->prior_slot = p->pose_cur;
    j.pose_from = p->pose_cur;
    j.pose_to = (p->pose_cur + 1) % 3;
    p->pose_cur = j.pose_to;
    p->mp_rows = 0;
    if (n_max > 0) {
        p->last_maps_prev = j.pd.maps_prev;
        p->last_cand = j.pd.cand;
    }
    return MV_OK;
}

// Device part: every launch of the frame's backend + solve.  Runs on the caller's thread, or on the launch thread (then it must
// not touch anything but the job, the Backend slot and the launch thread's own events / flags).
static int finish_issue(mvFramePipe* p, const FinishJob& j, const int64_t* perm_host) {
    const mvFramePipeConfig& c = p->c;
    const int L = p->lanes, cap = c.num_point > 0 ? c.num_point : 1;
    const int n_max = j.n_max;
    const Pending& pd = j.pd;
    const long g = j.g;
    const int k = (int)(g & 1);
    Backend& b = p->be[k];
    const int32_t* n_sel = j.n_sel;
    // front_on_decoder (device-driven frames of the alt layout): the pose-independent launch goes behind the frame's own selector segment on its decoder-side
    // stream — in order, so no barrier for the selector; consecutive frames' front launches (other decoder-side stream each) are chained by the ring event of the
    // previous finish, which also orders the generator's state buffers and covers the previous frame's segment
    const bool on_dec = j.device && p->front_on_decoder && p->alt && !p->sel_on_back;
    hipStream_t s = on_dec ? p->s_lk[j.pd.f % p->n_lk] : p->s_back;
    if (n_max == 0 && p->dev_draw)
        MV_HIP(hipMemcpyAsync(p->rp_state[(g + 1) & 1], p->rp_state[g & 1], (size_t)L * (size_t)mv_randperm_state_words() * sizeof(uint32_t),
                              hipMemcpyDeviceToDevice, s));
    if (n_max == 0) {   // nothing to track in any lane: the poses stay at the motion-model prior (MACVO.py:303-307)
        // The pose slots still rotate (MV_FB_POSE age a = the pose after the a-th newest finish) and the slot's events are
        // refreshed, so that everything keyed on "slot of finish g" (mv_frame_pipe_map_append, result views) sees this frame and
        // not the one two finishes back.  In-order on the side stream: behind the previous solve, no extra wait needed.
        MV_HIP(hipMemcpyAsync(p->pose[j.pose_to], p->pose[j.pose_from], (size_t)L * 7 * sizeof(float), hipMemcpyDeviceToDevice,
                              p->s_side));
        if (j.pose_sink)
            MV_HIP(hipMemcpyAsync(j.pose_sink, p->pose[j.pose_to], (size_t)L * 7 * sizeof(float), hipMemcpyDeviceToDevice, p->s_side));
        MV_HIP(hipEventRecord(p->e_posed[k], p->s_side));
        MV_HIP(hipEventRecord(p->e_pgo, p->s_side));
        MV_HIP(hipEventRecord(p->e_solved[k], p->s_side));
        MV_TRY(wait_if_pending(s, p->e_cand[pd.cand]));   // (lean chain: this frame's ring event also releases its volume buffer — not before its lookups are through)
        MV_HIP(hipEventRecord(p->e_backend[g % N_BEV], s));
        p->pgo_valid = true;
        p->solved_valid[k] = true;
        p->backend_valid[g % N_BEV] = true;
        p->nvalid_valid[k] = false;   // (mv_frame_pipe_wait_tracked then reports 0 observations: no mapping either)
        return MV_OK;
    }
    const Maps &m0 = p->maps[pd.maps_prev], &m1 = p->maps[pd.maps];

    // Slot reuse.  This backend slot's tables were last read by the solve of frame g - 2 (side stream) and by whoever looked at
    // that frame's result views: the former has its event, the latter the event of mv_frame_pipe_release (a consumer that
    // reads views asynchronously on its own stream calls it before it asks for the next frame).
    if (p->solved_valid[k] && s != p->s_side) MV_TRY(wait_if_pending(s, p->e_solved[k]));   // (alt layout: backend and solve share one in-order stream)
    if (p->release_valid) MV_TRY(wait_if_pending(s, p->e_release));
    // permutations [lanes, cap] -> pinned slot -> device (ONE copy; rows beyond a lane's n_sel are never read)
    const int ps = (int)(g % N_PERM);
    if (!j.device && p->dev_draw)   // a host-permuted finish in a device-driven pipe: the generators move on to the buffer the NEXT finish reads, unchanged
        MV_HIP(hipMemcpyAsync(p->rp_state[(g + 1) & 1], p->rp_state[g & 1], (size_t)L * (size_t)mv_randperm_state_words() * sizeof(uint32_t),
                              hipMemcpyDeviceToDevice, s));
    if (!j.device) {
        if (p->perm_valid[ps]) MV_HIP(hipEventSynchronize(p->e_perm[ps]));   // long done; keeps the slot reuse provably safe
        for (int l = 0; l < L; ++l)
            memcpy(p->h_perm[ps] + (size_t)l * cap, perm_host + (size_t)l * cap, (size_t)n_sel[l] * sizeof(int64_t));
    }
    if (!on_dec) MV_TRY(wait_if_pending(s, p->e_cand[pd.cand]));   // fired: the host has just read this frame's count (device-driven: a pending barrier)
    else if (g > 0 && p->backend_valid[(g - 1) % N_BEV]) MV_TRY(wait_if_pending(s, p->e_backend[(g - 1) % N_BEV]));
    // alt layout: the previous frame's maps (gathers below) were written by a segment on the other decoder-side stream, which e_cand does not cover.
    // (Slot f - 1 of e_seg is re-recorded by frame f - 1 + N_INEV, far beyond the frames in flight.)
    // ... unless the previous frame's backend ran on THIS stream and waited for that segment itself (its e_cand is recorded at the same point): stream order
    // then covers it, and a pending cross-queue barrier less sits in front of the backend (device-driven frames are issued long before their selector is done;
    // every unsatisfied barrier at the head of a queue costs the other queues dispatch latency, profiles/r06_device_draw_ab.log)
    const bool prev_here = p->alt && p->last_backend_frame == pd.f - 1;   // (on_dec: the previous front launch's event, waited for above, covers it too)
    if (p->alt && pd.f > 0 && !prev_here) MV_TRY(wait_if_pending(s, p->e_seg[(pd.f - 1) % N_INEV]));
    p->last_backend_frame = pd.f;
    const int ti = pd.ti;
    if (ti >= 0) MV_HIP(hipEventRecord(p->tv4[ti], s));
    const bool perm_in_args = L == 1 && n_max <= 256;   // one lane: the permutation rides in the kernel arguments (no pinned staging copy, no H2D node)
    if (!perm_in_args && !j.device) {
        const size_t perm_bytes = ((size_t)(L - 1) * cap + n_sel[L - 1]) * sizeof(int64_t);
        MV_HIP(hipMemcpyAsync(b.perm, p->h_perm[ps], perm_bytes, hipMemcpyHostToDevice, s));
        MV_HIP(hipEventRecord(p->e_perm[ps], s));
        p->perm_valid[ps] = true;
    }
    // Pose-INDEPENDENT part first: it overlaps the previous frames' solves.  The chain solve(t-1) -> backend(t) -> solve(t) is the
    // sequential dependency of visual odometry and, beside a GEMM that never pauses, it was the period of a single-sequence
    // stream (track 19 + back-projection 9 + covariances 48 + filters 21 + solve 108 us + launch gaps and two stream hops =
    // ~290 us).  Only the rotation into the world frame needs the previous pose (MACVO.py:273-281): it runs as one tiny kernel on
    // the SOLVE's stream right behind the previous solve, so the critical chain is solve -> mv_pose_apply_lanes -> solve on one
    // in-order stream.
    const bool fused = p->fuse_backend != 0;
    mvMatchCovParams cp{c.H, c.W, c.cov_kernel_size, 1, c.fx, c.fy, c.cx, c.cy, c.min_flow_cov_sq, c.min_depth_cov};
    if (j.device) {
        // device-driven frame: the front launch draws the permutation itself (count and generator in device memory) and publishes the live-row count
        MV_TRY(mv_backend_front_draw_lanes(p->cand[pd.cand], (size_t)p->plane, p->count[pd.cand], 4, p->rp_state[g & 1], p->rp_state[(g + 1) & 1], c.num_point,
                                           L, cap, m1.match_flow, m1.match_cov, m0.depth, m0.disparity, m0.disparity_cov, m0.depth_cov, m1.depth, m1.disparity,
                                           m1.disparity_cov, m1.depth_cov, c.edgewidth, c.match_cov_default, &cp, b.perm, b.live_dev, b.kp0, b.kp0f, b.kp1,
                                           b.inbound, b.vals, b.sigma0, b.sigma1, b.pos_Tc, b.cov0, b.cov1, s));
    } else if (fused) {
        // gather + track + back-projection + both covariance models + observation filters = ONE launch
        MV_TRY(mv_backend_front_lanes(p->cand[pd.cand], (size_t)p->plane, b.perm, perm_in_args ? p->h_perm[ps] : nullptr, L, n_sel, cap,
                                      m1.match_flow, m1.match_cov, m0.depth, m0.disparity, m0.disparity_cov, m0.depth_cov, m1.depth, m1.disparity,
                                      m1.disparity_cov, m1.depth_cov, c.edgewidth, c.match_cov_default, &cp, b.kp0, b.kp0f, b.kp1, b.inbound, b.vals,
                                      b.sigma0, b.sigma1, b.pos_Tc, b.cov0, b.cov1, s));
        // (the observation filters: prologue of the solve's launch below; mapping mode needs the count on the host first and keeps the launch)
        if (c.mapping)
            MV_TRY(mv_obs_filter_lanes(b.inbound, b.cov0, b.cov1, b.vals, c.filters, c.filter_min_depth, c.max_depth, L, n_sel, cap,
                                       b.valid, b.n_valid, s));
    } else {   // MV_PIPE_FUSE_BACKEND=0: the five-launch form (the reference point of the bitwise test)
        MV_TRY(mv_kp_front_lanes(p->cand[pd.cand], (size_t)p->plane, b.perm, perm_in_args ? p->h_perm[ps] : nullptr, L, n_sel, cap,
                                 m1.match_flow, m1.match_cov, m0.depth, m0.disparity, m0.disparity_cov, m0.depth_cov, m1.depth,
                                 m1.disparity, m1.disparity_cov, m1.depth_cov, c.H, c.W, c.edgewidth, c.match_cov_default, c.fx, c.fy,
                                 c.cx, c.cy, b.kp0, b.kp0f, b.kp1, b.inbound, b.vals, b.sigma0, b.sigma1, b.pos_Tc, s));
    }
    if (!fused) {
        MV_TRY(mv_match_cov_pair_lanes(m0.depth, b.kp0f, b.sigma0, nullptr, b.cov0, nullptr, m1.depth, b.kp1, b.sigma1, b.cov1, &cp,
                                       L, n_sel, cap, s));
        MV_TRY(mv_obs_filter_lanes(b.inbound, b.cov0, b.cov1, b.vals, c.filters, c.filter_min_depth, c.max_depth, L, n_sel, cap,
                                   b.valid, b.n_valid, s));
    }
    if (c.mapping) {   // the mapping decision of this frame (MACVO.py:303-307) needs the observation count on the host
        MV_HIP(hipMemcpyAsync(p->h_nvalid[k], b.n_valid, (size_t)L * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        MV_HIP(hipEventRecord(p->e_nvalid[k], s));
        p->nvalid_valid[k] = true;
    }
    MV_HIP(hipEventRecord(p->e_backend[g % N_BEV], s));
    p->backend_valid[g % N_BEV] = true;
    if (ti >= 0) MV_HIP(hipEventRecord(p->tv5[ti], s));

    // ---- side stream: world-frame tables from the previous solve's pose (same stream: no event), then the LM solves of all
    // lanes in ONE launch (problem l = rows [l * cap, (l + 1) * cap), dead rows masked by `valid`); the optimised poses become
    // the next frame's priors (StaticMotionModel)
    hipStream_t ss = p->s_side;
    if (ss != s) MV_HIP(hipStreamWaitEvent(ss, p->e_backend[g % N_BEV], 0));   // (alt layout: backend + solve share one in-order stream)
    const float* pose = p->pose[j.pose_from];
    if (ti >= 0) MV_HIP(hipEventRecord(p->tv6[ti], ss));
    const size_t N = (size_t)cap;
    const size_t LN = (size_t)L * N;   // value table is [11, lanes, cap]: each of its rows is one concatenated per-point column
    if (j.device) {
        MV_TRY(mv_pgo_solve_posed_dev(L, p->offs, b.live_dev, 2, cap, c.graph_type, pose, p->intr, p->bl, b.pos_Tc, b.cov0, b.pos_Tw, b.cov0w, b.rot, b.kp1,
                                      b.vals + 4 * LN, b.vals + 5 * LN, b.vals + 6 * LN, b.sigma1, b.cov1, c.filters, c.filter_min_depth, c.max_depth, b.inbound,
                                      b.vals, b.valid, b.n_valid, c.min_num_point, &c.lm, b.pose64, b.info, p->pose[j.pose_to], j.pose_sink, ss));
        MV_HIP(hipEventRecord(p->e_solved[k], ss));
        MV_HIP(hipEventRecord(p->e_pgo, ss));
    } else if (p->fuse_backend) {
        // ... and the pose-dependent half = ONE launch: the rotation into the world frame is the solve kernel's prologue, the caller's pose
        // sink its second output (the 28-byte device-to-device copy was a DMA node on the critical stream).  No e_posed: the world-frame
        // tables are consumed behind e_solved (mv_frame_pipe_map_append).
        MV_TRY(mv_pgo_solve_posed(L, p->offs, n_sel, cap, c.graph_type, pose, p->intr, p->bl, b.pos_Tc, b.cov0, b.pos_Tw, b.cov0w, b.rot, b.kp1,
                                  b.vals + 4 * LN, b.vals + 5 * LN, b.vals + 6 * LN, b.sigma1, b.cov1, c.mapping ? -1 : c.filters, c.filter_min_depth,
                                  c.max_depth, b.inbound, b.vals, b.valid, b.n_valid, c.min_num_point, &c.lm, b.pose64, b.info,
                                  p->pose[j.pose_to], j.pose_sink, ss));
        MV_HIP(hipEventRecord(p->e_solved[k], ss));
        MV_HIP(hipEventRecord(p->e_pgo, ss));
    } else {
        MV_TRY(mv_pose_apply_lanes(pose, b.pos_Tc, b.cov0, L, n_sel, cap, b.pos_Tw, b.rot, b.cov0w, ss));
        MV_HIP(hipEventRecord(p->e_posed[k], ss));
        MV_TRY(mv_pgo_solve(L, p->offs, c.graph_type, pose, p->intr, p->bl, b.pos_Tw, b.cov0w, b.kp1, b.vals + 4 * LN,
                            b.vals + 5 * LN, b.vals + 6 * LN, b.sigma1, b.cov1, b.valid, c.min_num_point, &c.lm, b.pose64, b.info,
                            p->pose[j.pose_to], ss));
        if (j.pose_sink)
            MV_HIP(hipMemcpyAsync(j.pose_sink, p->pose[j.pose_to], (size_t)L * 7 * sizeof(float), hipMemcpyDeviceToDevice, ss));
        MV_HIP(hipEventRecord(p->e_pgo, ss));
        MV_HIP(hipEventRecord(p->e_solved[k], ss));
    }
    if (ti >= 0) MV_HIP(hipEventRecord(p->tv7[ti], ss));
    p->pgo_valid = true;
    p->solved_valid[k] = true;
    return MV_OK;
}

// permutations of a seeded job: torch.randperm of each lane's generator (see above)
static const int64_t* draw_perms(mvFramePipe* p, const FinishJob& j) {
    const int cap = p->c.num_point > 0 ? p->c.num_point : 1;
    for (int l = 0; l < p->lanes; ++l)
        randperm_head(p->rng[(size_t)l], j.n_cand[l], p->c.num_point, p->perm_scratch, p->perm_host.data() + (size_t)l * cap);
    return p->perm_host.data();
}

// ------------------------------------------------------------------------------------------------ backend launch thread

static void launch_thread_main(mvFramePipe* p) {
    (void)hipSetDevice(p->device);
    long taken = 0;
    for (;;) {
        FinishJob j;
        // A job arrives once per frame period (~150 us) and sits on the frame's critical chain (selector count -> backend): waking a thread that
        // sleeps on a condition variable costs 30-60 us of it.  Spin on the submission counter for up to MV spin_us first (one period and a half),
        // sleep only when the stream has really gone quiet.
        if (p->spin_us > 0) {
            const double t_end = now_us() + p->spin_us;
            int it = 0;
            while (p->n_submitted.load(std::memory_order_acquire) == taken && !p->stop_flag.load(std::memory_order_relaxed)) {
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#else
                std::this_thread::yield();
#endif
                if ((++it & 255) == 0 && now_us() > t_end) break;
            }
        }
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv_job.wait(lk, [&] { return p->stop || !p->jobs.empty(); });
            if (p->jobs.empty()) return;
            j = std::move(p->jobs.front());
            p->jobs.pop_front();
        }
        ++taken;
        const double t_pick = now_us();
        if (p->host_stats) { p->st_n += 1; p->st_submit_to_pick += t_pick - j.t_submit; p->st_count_to_submit += j.t_submit - j.t_count; }
        int rc = finish_issue(p, j, j.seeded ? draw_perms(p, j) : j.perm.data());
        if (p->host_stats) p->st_pick_to_issued += now_us() - t_pick;
        if (rc == MV_OK && j.has_sel) rc = issue_selector_segment(p, j.sel);
        {
            std::lock_guard<std::mutex> lk(p->mu);
            p->issued = j.g + 1;
            if (rc != MV_OK && p->async_rc == MV_OK) p->async_rc = rc;
        }
        p->cv_done.notify_all();
    }
}

// block until the launch thread has issued the first `n` finishes (n <= n_fin); returns the first asynchronous error
static int wait_issued(mvFramePipe* p, long n) {
    if (!p->async_backend) return MV_OK;
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv_done.wait(lk, [&] { return p->issued >= n; });
    return p->async_rc;
}
static int flush_jobs(mvFramePipe* p) { return wait_issued(p, p->n_fin); }

static int submit_or_issue(mvFramePipe* p, FinishJob& j, const int64_t* perm_host) {
    if (!p->async_backend) {
        MV_TRY(finish_issue(p, j, j.seeded ? draw_perms(p, j) : perm_host));
        return j.has_sel ? issue_selector_segment(p, j.sel) : MV_OK;
    }
    if (!j.seeded && !j.device && j.n_max > 0) {
        const size_t cap = p->c.num_point > 0 ? p->c.num_point : 1;
        j.perm.assign(perm_host, perm_host + (size_t)p->lanes * cap);
    }
    int rc;
    j.t_submit = now_us();
    {
        std::lock_guard<std::mutex> lk(p->mu);
        rc = p->async_rc;
        p->jobs.push_back(std::move(j));
    }
    p->n_submitted.fetch_add(1, std::memory_order_release);
    p->cv_job.notify_one();
    return rc;
}

// wait_candidates + permutations (per-lane generators of mv_frame_pipe_seed_lanes) + finish in one host call
extern "C" int mv_frame_pipe_finish_seeded(mvFramePipe* p, float* pose_sink, int32_t* n_cand_out, int32_t* n_sel_out) {
    MV_CHECK_ARG(p && !p->pending.empty() && (int)p->rng.size() == p->lanes);
    MV_CHECK_ARG(!p->dev_draw);   // (its generators live on the device: mv_frame_pipe_finish_device)
    MV_TRY(selector_of_front_issued(p));
    const Pending& pd = p->pending.front();
    const double t_w0 = p->host_stats ? now_us() : 0.0;
    MV_HIP(hipEventSynchronize(p->e_cand[pd.cand]));
    FinishJob j{};
    j.seeded = true;
    j.t_count = now_us();
    if (p->host_stats) p->st_wait += j.t_count - t_w0;
    int32_t nsel[MV_MAX_LANES];
    for (int l = 0; l < p->lanes; ++l) {
        const int64_t n = p->h_count[pd.cand][4 * l];
        const int k = (int)(n < p->c.num_point ? n : p->c.num_point);
        j.n_cand[l] = n;
        nsel[l] = k;
        if (n_cand_out) n_cand_out[l] = (int32_t)n;
        if (n_sel_out) n_sel_out[l] = k;
    }
    MV_TRY(finish_host(p, nsel, pose_sink, j));
    return submit_or_issue(p, j, nullptr);
}

// The device-driven finish (round 6): nothing to wait for and nothing to draw — the frame's backend + solve are queued behind its selector segment by event, the
// permutation head is drawn inside the front launch.  The host learns the counts only if it asks (mv_frame_pipe_finished_counts).
extern "C" int mv_frame_pipe_finish_device(mvFramePipe* p, float* pose_sink) {
    MV_CHECK_ARG(p && !p->pending.empty() && p->dev_draw);
    MV_TRY(selector_of_front_issued(p));
    FinishJob j{};
    j.seeded = false;
    j.device = true;
    j.t_count = j.t_submit = p->host_stats ? now_us() : 0.0;
    int32_t nsel[MV_MAX_LANES];
    for (int l = 0; l < p->lanes; ++l) nsel[l] = p->c.num_point;   // upper bound: rows beyond the live ones are masked by `valid`
    MV_TRY(finish_host(p, nsel, pose_sink, j));
    const int rc = submit_or_issue(p, j, nullptr);
    if (p->host_stats) { p->st_fin += now_us() - j.t_count; p->st_calls += 1; }
    return rc;
}

// candidate / selected-keypoint counts of the `age`-th newest FINISHED frame (age 0 or 1), read back from the frame's backend slot: blocks until that
// frame's front launch has run.  Off the hot path (result views, tests).
extern "C" int mv_frame_pipe_finished_counts(mvFramePipe* p, int age, int32_t* n_cand, int32_t* n_sel) {
    MV_CHECK_ARG(p && p->dev_draw && age >= 0 && age <= 1 && p->n_fin - 1 - age >= 0);
    MV_TRY(flush_jobs(p));
    const long g = p->n_fin - 1 - age;
    MV_HIP(hipEventSynchronize(p->e_backend[g % N_BEV]));
    std::vector<int32_t> h(2 * (size_t)p->lanes);
    MV_HIP(hipMemcpy(h.data(), p->be[g & 1].live_dev, h.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int l = 0; l < p->lanes; ++l) {
        if (n_sel) n_sel[l] = h[2 * l];
        if (n_cand) n_cand[l] = h[2 * l + 1];
    }
    return MV_OK;
}

// Bound how far the host runs ahead of the GPU in a device-driven stream: blocks until the front launch of the finish `lag` finishes back (0 = the newest) has run.
extern "C" int mv_frame_pipe_wait_finished(mvFramePipe* p, int lag) {
    MV_CHECK_ARG(p && lag >= 0 && lag < N_BEV - 1);
    const long g = p->n_fin - 1 - lag;
    if (g < 0) return MV_OK;
    MV_TRY(wait_issued(p, g + 1));
    // (polling with hipEventQuery before blocking was measured: no difference — the wait is 10-40 us per frame and two frames back, profiles/r06_chain_gaps.log)
    MV_HIP(hipEventSynchronize(p->e_backend[g % N_BEV]));
    return MV_OK;
}

// Host-only probe of the driver's permutation generator (no GPU, no pipe): `calls` successive `torch.randperm(n[i])[:k]` of one
// generator seeded with `seed`, written to out[i * k .. i * k + min(k, n[i])) — what mv_frame_pipe_finish_seeded draws for one lane
// frame after frame.  Lets the CPU test suite pin the generator against torch itself.
extern "C" int mv_randperm_heads(uint64_t seed, const int64_t* n, int calls, int k, int64_t* out) {
    MV_CHECK_ARG(n && out && calls >= 0 && k >= 0);
    std::mt19937 eng((uint32_t)(seed & 0xffffffffull));
    std::vector<int32_t> scratch;
    for (int i = 0; i < calls; ++i) {
        MV_CHECK_ARG(n[i] >= 0 && n[i] < ((int64_t)1 << 31));
        randperm_head(eng, n[i], k, scratch, out + (size_t)i * k);
    }
    return MV_OK;
}

extern "C" int mv_frame_pipe_finish(mvFramePipe* p, const int64_t* perm_host, const int32_t* n_sel, float* pose_sink) {
    MV_CHECK_ARG(p && n_sel && !p->pending.empty());
    for (int l = 0; l < p->lanes; ++l)
        MV_CHECK_ARG(n_sel[l] >= 0 && n_sel[l] <= p->c.num_point && (n_sel[l] == 0 || perm_host));
    MV_TRY(selector_of_front_issued(p));
    MV_TRY(ensure_chain_events(p, p->pending.front().f, p->pending.front().cand));       // (lean chain: a host-permuted finish waits for the markers)
    if (p->pending.front().f > 0) MV_TRY(ensure_chain_events(p, p->pending.front().f - 1, -1));
    FinishJob j{};
    j.seeded = false;
    MV_TRY(finish_host(p, n_sel, pose_sink, j));
    return submit_or_issue(p, j, perm_host);
}

// Register the newest FINISHED frame in a device-resident map (call right after mv_frame_pipe_finish; lanes == 1): the
// frame's tables are handed to mv_map_append where they lie (no copies), on the backend stream behind the observation filter;
// the optimised pose is then written over the frame's prior on the solve stream (write_graph_data, Optimizer.py:104-108).
extern "C" int mv_frame_pipe_map_append(mvFramePipe* p, const mvMapStores* stores, int frame_idx, int prev_frame,
                                        const float* K_dev, const float* T_BS_dev, float baseline, int64_t time_ns,
                                        const uint8_t* color_dev) {
    MV_CHECK_ARG(p && stores && K_dev && T_BS_dev && p->lanes == 1 && p->n_fin > 0 && frame_idx >= 0);
    MV_TRY(flush_jobs(p));
    const mvFramePipeConfig& c = p->c;
    const long g = p->n_fin - 1;
    const Backend& b = p->be[g & 1];
    const int cap = c.num_point > 0 ? c.num_point : 1;
    mvMapFrame f{};
    f.n_rows = b.n_sel[0];
    f.table_stride = cap;
    f.prev_frame = prev_frame;
    f.min_num_point = c.min_num_point;
    f.valid = b.valid;
    f.kp0 = b.kp0f; f.kp1 = b.kp1; f.vals = b.vals; f.sigma0 = b.sigma0; f.sigma1 = b.sigma1;
    f.cov0 = b.cov0; f.cov1 = b.cov1; f.pos_Tw = b.pos_Tw; f.cov0_world = b.cov0w;
    f.color = color_dev;
    f.K = K_dev; f.T_BS = T_BS_dev;
    f.prior_pose = p->pose[p->prior_slot];
    f.baseline = baseline;
    f.time_ns = time_ns;
    f.out_frame_idx = nullptr;
    MV_HIP(hipStreamWaitEvent(p->s_back, p->fuse_backend ? p->e_solved[g & 1] : p->e_posed[g & 1], 0));   // pos_Tw / cov0_world come from the side stream
    MV_TRY(mv_map_append(&f, stores, p->s_back));
    MV_HIP(hipEventRecord(p->e_map, p->s_back));
    // the backend tables must outlive the append: later backends run on the same stream (ordered); the optimised pose goes
    // in after the append wrote the prior
    MV_HIP(hipStreamWaitEvent(p->s_side, p->e_map, 0));
    MV_HIP(hipMemcpyAsync(stores->pose + 7 * (size_t)frame_idx, p->pose[p->pose_cur], 7 * sizeof(float), hipMemcpyDeviceToDevice,
                          p->s_side));
    MV_HIP(hipEventRecord(p->e_pgo, p->s_side));
    p->pgo_valid = true;
    return MV_OK;
}

// ------------------------------------------------------------------------------------------------ dense-mapping tail
extern "C" int mv_frame_pipe_wait_tracked(mvFramePipe* p, int32_t* n_valid, int32_t* n_cand_map) {
    MV_CHECK_ARG(p && n_valid && p->c.mapping && p->n_fin > 0);
    MV_TRY(flush_jobs(p));
    const int k = (int)((p->n_fin - 1) & 1);
    if (!p->nvalid_valid[k]) {          // a frame without keypoints: nothing was tracked
        *n_valid = 0;
        if (n_cand_map) *n_cand_map = 0;
        return MV_OK;
    }
    MV_HIP(hipEventSynchronize(p->e_nvalid[k]));
    *n_valid = p->h_nvalid[k][0];
    if (n_cand_map) *n_cand_map = p->h_count_m[p->last_cand][0];   // (its copy preceded e_cand, which the host waited for before finish)
    return MV_OK;
}

extern "C" int mv_frame_pipe_map_points(mvFramePipe* p, const int64_t* perm_host, int n_sel, const float* image_dev,
                                        const mvMapStores* stores) {
    MV_CHECK_ARG(p && p->c.mapping && p->n_fin > 0 && n_sel >= 0 && n_sel <= p->c.map_num_point && (n_sel == 0 || perm_host));
    MV_TRY(flush_jobs(p));
    const mvFramePipeConfig& c = p->c;
    hipStream_t s = p->s_back;
    const Maps& m0 = p->maps[p->last_maps_prev];
    if (p->maptail_valid) MV_HIP(hipEventSynchronize(p->e_maptail));   // pinned permutation + map buffers of the previous tail (long done)
    if (p->release_valid) MV_TRY(wait_if_pending(s, p->e_release));    // ... and consumers of its result views
    if (n_sel > 0) {
        memcpy(p->h_perm_m, perm_host, (size_t)n_sel * sizeof(int64_t));
        MV_HIP(hipMemcpyAsync(p->mp_perm, p->h_perm_m, (size_t)n_sel * sizeof(int64_t), hipMemcpyHostToDevice, s));
        MV_TRY(mv_kp_gather(p->cand_m[p->last_cand], p->mp_perm, n_sel, c.W, p->mp_uv, s));
        // prev_pose.Act (MACVO.py:334): the pose the frame started from = the previous frame's optimised pose
        MV_TRY(wait_if_pending(s, p->e_pgo));   // (the solve that produced it; the newest solve is recorded later on `side`: also fine)
        MV_TRY(mv_map_points(p->mp_uv, n_sel, m0.depth, m0.depth_cov, image_dev, c.H, c.W, c.fx, c.fy, c.cx, c.cy,
                             p->pose[p->prior_slot], c.match_cov_default, p->mp_uvf, p->mp_d, p->mp_sdd, p->mp_sigma, p->mp_Tc, p->mp_Tw,
                             image_dev ? p->mp_color : nullptr, s));
        mvMatchCovParams cp{c.H, c.W, c.cov_kernel_size, 1, c.fx, c.fy, c.cx, c.cy, c.min_flow_cov_sq, c.min_depth_cov};
        MV_TRY(mv_match_cov(m0.depth, p->mp_uvf, p->mp_sigma, p->mp_sdd, nullptr, &cp, n_sel, p->mp_cov, nullptr, nullptr, s));
    }
    if (stores) {   // map_points.push + frame2map.add (also for n_sel == 0: the reference adds the empty range), behind the frame's registration
        MV_TRY(mv_map_append_points(stores, n_sel, p->mp_Tw, p->mp_cov, image_dev ? p->mp_color : nullptr, s));
    }
    p->mp_rows = n_sel;
    MV_HIP(hipEventRecord(p->e_maptail, s));
    p->maptail_valid = true;
    MV_HIP(hipEventRecord(p->e_backend[(p->n_fin - 1) % N_BEV], s));   // the previous frame's maps stay in use until here (the next enqueue waits for this event)
    return MV_OK;
}

extern "C" int mv_frame_pipe_release(mvFramePipe* p, mvStream_t stream) {
    MV_CHECK_ARG(p);
    MV_HIP(hipEventRecord(p->e_release, (hipStream_t)stream));
    p->release_valid = true;
    return MV_OK;
}

extern "C" int mv_frame_pipe_sync(mvFramePipe* p, mvStream_t stream, int block_host) {
    MV_CHECK_ARG(p);
    MV_TRY(flush_deferred(p));
    MV_TRY(flush_jobs(p));   // everything finished so far has been issued
    if (block_host == 2) {   // results of the newest FINISHED frame only: its solve (which ran behind its backend kernels)
        if (p->pgo_valid) MV_HIP(hipStreamWaitEvent((hipStream_t)stream, p->e_pgo, 0));
        return MV_OK;
    }
    if (block_host) {
        MV_HIP(hipStreamSynchronize(p->s_vol));
        for (int k = 0; k < p->n_lk; ++k) MV_HIP(hipStreamSynchronize(p->s_lk[k]));
        MV_HIP(hipStreamSynchronize(p->s_back));
        MV_HIP(hipStreamSynchronize(p->s_side));
        if (p->s_sel) MV_HIP(hipStreamSynchronize(p->s_sel));
        return MV_OK;
    }
    // make `stream` wait for everything enqueued so far
    hipEvent_t e;
    MV_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipStream_t all[4 + MAX_LK] = {p->s_vol, p->s_back, p->s_side, p->s_sel};
    for (int k = 0; k < p->n_lk; ++k) all[4 + k] = p->s_lk[k];
    int rc = MV_OK;
    for (hipStream_t q : all) {
        if (!q) continue;
        if (hipEventRecord(e, q) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, e, 0) != hipSuccess) rc = MV_ERR_LAUNCH;
    }
    (void)hipEventDestroy(e);
    return rc;
}

extern "C" int mv_frame_pipe_time_volume(mvFramePipe* p, int max_launches) {
    MV_CHECK_ARG(p && max_launches >= 0 && max_launches <= (1 << 20));
    MV_TRY(flush_jobs(p));   // finish_issue on the launch thread records into tv3..tv7 (which may reallocate below)
    while ((int)p->tv0.size() < max_launches) {
        hipEvent_t a, b;
        MV_HIP(hipEventCreate(&a));
        p->tv0.push_back(a);
        MV_HIP(hipEventCreate(&b));
        p->tv1.push_back(b);
        hipEvent_t c2, c3;
        MV_HIP(hipEventCreate(&c2));
        p->tv2.push_back(c2);
        MV_HIP(hipEventCreate(&c3));
        p->tv3.push_back(c3);
        for (auto* v : {&p->tv4, &p->tv5, &p->tv6, &p->tv7}) {
            hipEvent_t e;
            MV_HIP(hipEventCreate(&e));
            v->push_back(e);
        }
    }
    p->n_timed = 0;
    p->timed_cap = max_launches;
    return MV_OK;
}

extern "C" int mv_frame_pipe_time_detail(mvFramePipe* p, int on) {
    MV_CHECK_ARG(p);
    MV_TRY(flush_jobs(p));
    p->time_detail = on ? 1 : 0;
    return MV_OK;
}

extern "C" int mv_frame_pipe_volume_times(mvFramePipe* p, float* ms, int cap, int* n) {
    MV_CHECK_ARG(p && n && cap >= 0 && (cap == 0 || ms));
    MV_TRY(flush_jobs(p));
    MV_HIP(hipStreamSynchronize(p->s_vol));
    const int m = p->n_timed < cap ? p->n_timed : cap;
    for (int i = 0; i < m; ++i) MV_HIP(hipEventElapsedTime(&ms[i], p->tv0[i], p->tv1[i]));
    *n = m;
    return MV_OK;
}

// ... and when each timed GEMM STARTED, ms since the first one (the steady-state period of the very pass whose wall time is the bench line's `value`:
// only the event pairs around the GEMM are recorded in that pass)
extern "C" int mv_frame_pipe_volume_starts(mvFramePipe* p, float* ms, int cap, int* n) {
    MV_CHECK_ARG(p && n && cap >= 0 && (cap == 0 || ms));
    MV_TRY(flush_jobs(p));
    MV_HIP(hipStreamSynchronize(p->s_vol));
    const int m = p->n_timed < cap ? p->n_timed : cap;
    for (int i = 0; i < m; ++i) MV_HIP(hipEventElapsedTime(&ms[i], p->tv0[0], p->tv0[i]));
    *n = m;
    return MV_OK;
}

// timeline of the timed frames: for frame i, ms[4*i + {0,1,2,3}] = GEMM start, GEMM end, last lookup done, selector done,
// all relative to the first timed GEMM start (blocks until everything enqueued so far has finished)
extern "C" int mv_frame_pipe_timeline(mvFramePipe* p, float* ms, int cap_frames, int* n) {
    MV_CHECK_ARG(p && n && cap_frames >= 0 && (cap_frames == 0 || ms));
    MV_TRY(flush_jobs(p));   // in the 'late' selector placement tv3 is recorded on the launch thread
    if (!p->time_detail) { *n = 0; return MV_OK; }   // only the GEMM pairs were recorded
    MV_HIP(hipStreamSynchronize(p->s_vol));
    for (int k = 0; k < p->n_lk; ++k) MV_HIP(hipStreamSynchronize(p->s_lk[k]));
    MV_HIP(hipStreamSynchronize(p->s_back));
    if (p->s_sel) MV_HIP(hipStreamSynchronize(p->s_sel));
    const int m = p->n_timed < cap_frames ? p->n_timed : cap_frames;
    for (int i = 0; i < m; ++i) {
        hipEvent_t evs[4] = {p->tv0[i], p->tv1[i], p->tv2[i], p->tv3[i]};
        for (int j = 0; j < 4; ++j)
            if (hipEventElapsedTime(&ms[4 * i + j], p->tv0[0], evs[j]) != hipSuccess) ms[4 * i + j] = -1.f;
    }
    *n = m;
    return MV_OK;
}

// ... and the backend side of the same frames: ms[4*i + {0,1,2,3}] = backend start, backend end (backend stream), pose_apply start,
// solve end (solve stream); -1 where a frame was not finished (or had no keypoints)
extern "C" int mv_frame_pipe_timeline_backend(mvFramePipe* p, float* ms, int cap_frames, int* n) {
    MV_CHECK_ARG(p && n && cap_frames >= 0 && (cap_frames == 0 || ms));
    MV_TRY(flush_jobs(p));
    if (!p->time_detail) { *n = 0; return MV_OK; }
    MV_HIP(hipStreamSynchronize(p->s_back));
    MV_HIP(hipStreamSynchronize(p->s_side));
    for (int k = 0; k < p->n_lk; ++k) MV_HIP(hipStreamSynchronize(p->s_lk[k]));   // (device-driven frames run the front launch on the decoder side)
    const int m = p->n_timed < cap_frames ? p->n_timed : cap_frames;
    for (int i = 0; i < m; ++i) {
        hipEvent_t evs[4] = {p->tv4[i], p->tv5[i], p->tv6[i], p->tv7[i]};
        for (int j = 0; j < 4; ++j)
            if (hipEventElapsedTime(&ms[4 * i + j], p->tv0[0], evs[j]) != hipSuccess) ms[4 * i + j] = -1.f;
    }
    *n = m;
    return MV_OK;
}

extern "C" int mv_frame_pipe_buffer(mvFramePipe* p, int which, int age, void** ptr, size_t* count) {
    MV_CHECK_ARG(p && ptr && count && age >= 0);
    const mvFramePipeConfig& c = p->c;
    const size_t L = p->lanes, plane = L * p->plane;   // every buffer has a leading [lanes] dimension (VALS: [11, lanes, cap])
    *ptr = nullptr;
    *count = 0;
    // frontend-side buffers: age 0 = the newest enqueued frame; backend-side: age 0 = the newest finished frame
    const long f = p->n_enq - 1 - age, g = p->n_fin - 1 - age;
    auto front = [&](int depth) { return f >= 0 && age < depth; };
    auto back = [&]() { return g >= 0 && age < 2; };
    const Backend* b = back() ? &p->be[g & 1] : nullptr;
    const size_t N = L * (size_t)(c.num_point > 0 ? c.num_point : 1);   // capacity rows (a lane's live rows: its n_sel)
    switch (which) {
        case MV_FB_VOLUME:
            // age limit: with a GEMM issued ahead (n_vol > n_enq) the buffer of frame f - 1 is being rewritten
            if (!front(p->n_vol > p->n_enq ? 1 : 2)) break;
            *ptr = p->vol[f % p->n_volbuf]; *count = (size_t)c.pairs * p->n8 * p->n2t; return MV_OK;   // (n2t cells per slice: padded for a tiled fp16 volume with h8 % 4 != 0)
        case MV_FB_TOKENS: if (!front(1) || c.iters == 0) break; *ptr = p->tok[2 * (f % p->n_lk) + ((c.iters - 1) & 1)]; *count = (size_t)c.pairs * p->KK * p->n8; return MV_OK;
        case MV_FB_DISPARITY: if (!front(N_MAPS)) break; *ptr = p->maps[f % N_MAPS].disparity; *count = plane; return MV_OK;
        case MV_FB_DISPARITY_COV: if (!front(N_MAPS)) break; *ptr = p->maps[f % N_MAPS].disparity_cov; *count = plane; return MV_OK;
        case MV_FB_DEPTH: if (!front(N_MAPS)) break; *ptr = p->maps[f % N_MAPS].depth; *count = plane; return MV_OK;
        case MV_FB_DEPTH_COV: if (!front(N_MAPS)) break; *ptr = p->maps[f % N_MAPS].depth_cov; *count = plane; return MV_OK;
        case MV_FB_MATCH_FLOW: if (!front(N_MAPS)) break; *ptr = p->maps[f % N_MAPS].match_flow; *count = 2 * plane; return MV_OK;
        case MV_FB_MATCH_COV: if (!front(N_MAPS)) break; *ptr = p->maps[f % N_MAPS].match_cov; *count = 3 * plane; return MV_OK;
        case MV_FB_CAND: if (!front(N_CAND)) break; *ptr = p->cand[f % N_CAND]; *count = plane; return MV_OK;
        case MV_FB_COUNT: if (!front(N_CAND)) break; *ptr = p->count[f % N_CAND]; *count = 4 * L; return MV_OK;
        case MV_FB_STATS: if (!front(N_CAND)) break; *ptr = p->stats[f % N_CAND]; *count = 4 * L; return MV_OK;
        case MV_FB_KP0: if (!b) break; *ptr = b->kp0; *count = 2 * N; return MV_OK;
        case MV_FB_KP0F: if (!b) break; *ptr = b->kp0f; *count = 2 * N; return MV_OK;
        case MV_FB_KP1: if (!b) break; *ptr = b->kp1; *count = 2 * N; return MV_OK;
        case MV_FB_INBOUND: if (!b) break; *ptr = b->inbound; *count = N; return MV_OK;
        case MV_FB_VALS: if (!b) break; *ptr = b->vals; *count = 11 * N; return MV_OK;
        case MV_FB_SIGMA0: if (!b) break; *ptr = b->sigma0; *count = 3 * N; return MV_OK;
        case MV_FB_SIGMA1: if (!b) break; *ptr = b->sigma1; *count = 3 * N; return MV_OK;
        case MV_FB_POS_TC: if (!b) break; *ptr = b->pos_Tc; *count = 3 * N; return MV_OK;
        case MV_FB_POS_TW: if (!b) break; *ptr = b->pos_Tw; *count = 3 * N; return MV_OK;
        case MV_FB_ROT: if (!b) break; *ptr = b->rot; *count = 9 * L; return MV_OK;
        case MV_FB_COV0: if (!b) break; *ptr = b->cov0; *count = 9 * N; return MV_OK;
        case MV_FB_COV0W: if (!b) break; *ptr = b->cov0w; *count = 9 * N; return MV_OK;
        case MV_FB_COV1: if (!b) break; *ptr = b->cov1; *count = 9 * N; return MV_OK;
        case MV_FB_VALID: if (!b) break; *ptr = b->valid; *count = N; return MV_OK;
        case MV_FB_NVALID: if (!b) break; *ptr = b->n_valid; *count = L; return MV_OK;
        case MV_FB_POSE64: if (!b) break; *ptr = b->pose64; *count = 7 * L; return MV_OK;
        case MV_FB_INFO: if (!b) break; *ptr = b->info; *count = 4 * L; return MV_OK;
        case MV_FB_POSE: if (age > 1) break; *ptr = p->pose[(p->pose_cur + 3 - age) % 3]; *count = 7 * L; return MV_OK;
        case MV_FB_PERM: if (!b) break; *ptr = b->perm; *count = N; return MV_OK;
        case MV_FB_LIVE: if (!b || !p->dev_draw) break; *ptr = b->live_dev; *count = 2 * L; return MV_OK;
        case MV_FB_MAP_UV: if (!c.mapping || age) break; *ptr = p->mp_uvf; *count = 2 * (size_t)p->mp_rows; return MV_OK;
        case MV_FB_MAP_D: if (!c.mapping || age) break; *ptr = p->mp_d; *count = (size_t)p->mp_rows; return MV_OK;
        case MV_FB_MAP_SDD: if (!c.mapping || age) break; *ptr = p->mp_sdd; *count = (size_t)p->mp_rows; return MV_OK;
        case MV_FB_MAP_TC: if (!c.mapping || age) break; *ptr = p->mp_Tc; *count = 3 * (size_t)p->mp_rows; return MV_OK;
        case MV_FB_MAP_TW: if (!c.mapping || age) break; *ptr = p->mp_Tw; *count = 3 * (size_t)p->mp_rows; return MV_OK;
        case MV_FB_MAP_COV: if (!c.mapping || age) break; *ptr = p->mp_cov; *count = 9 * (size_t)p->mp_rows; return MV_OK;
        case MV_FB_MAP_COLOR: if (!c.mapping || age) break; *ptr = p->mp_color; *count = 3 * (size_t)p->mp_rows; return MV_OK;
        default: break;
    }
    return MV_ERR_INVALID_ARG;
}
