// PROBE (round 4; ran once on the MI355X with the round's last GPU seconds: tokens bit-identical, 121-127 us vs 111 us for the shipped kernel at B = 64,
// profiles/r04_lookup_persist_probe.log — slower as hipcc compiles it): a PERSISTENT, software-pipelined form of the batched 9x9 window lookup (VERDICT r3 next #4, configs[4]).
//
// Why.  The committed PMC pass of the batched lookup (profiles/r01_cfg4_b64_pmc_lookup_raw.txt: FETCH 271 MB at B = 64 before the margin trimming
// of round 3, i.e. ~205 MB now) + 100 MB of token writes in 116 us is ~2.6 TB/s of ACTUAL traffic — not the ~4.3 TB/s DESIGN §8 [r4] states
// (that figure multiplied the read inflation onto the writes as well): the kernel is latency-bound, not bandwidth-bound.  corr_lookup_kernel
// is one 32-query group per workgroup: coordinates -> addresses -> cell loads -> LDS -> taps -> LDS transpose -> stores, three barriers, and
// nothing of the next group in flight while the taps and the transpose run.  Here a workgroup walks groups g, g + G, g + 2 G, ..: the cell
// loads of group g + G are issued (into a second register set) and the coordinates of group g + 2 G requested BEFORE group g is processed.
// Arithmetic, LDS layout and the order of every operation of a group are corr_lookup_kernel's: the tokens must come out bit-identical
// (the .py harness checks that against mv_corr_lookup before it prints a time).
#include "../../mac-vo_amd/csrc/corr_lookup.hip"

namespace {

template <int R, int QPW, int QPB, typename VT = float>
__global__ __launch_bounds__(64 * (QPB / QPW)) void corr_lookup_persist_kernel(const VT* __restrict__ vol, const float* __restrict__ coords,
                                                                                float* __restrict__ out, int N1, int H2, int W2, int gpb /* groups per pair */,
                                                                                int total /* = B * gpb */, const VT* __restrict__ zero /* one readable 0 */) {
    constexpr int K = 2 * R + 1;
    constexpr int KK = K * K;
    constexpr int BS = K + 3;
    constexpr int CELLS = BS * BS;
    constexpr int NWAVE = QPB / QPW;
    constexpr int NTHR = 64 * NWAVE;
    constexpr int RPR = 64 / BS;
    constexpr int LPR = RPR * BS;
    constexpr int NROUND = (BS + RPR - 1) / RPR;
    static_assert((BS - 2) % RPR == 0, "the trimmed staging of corr_lookup_kernel (r = 4)");
    constexpr int NINNER = (BS - 2) / RPR;
    constexpr int TAP_ROUNDS = (KK + 63) / 64;
    constexpr int QPA = 64 / (2 * K);
    constexpr int APASS = (QPW + QPA - 1) / QPA;
    constexpr int BLK_STRIDE = QPW * CELLS;
    constexpr int BLK_FLOATS = NWAVE * BLK_STRIDE;
    constexpr int OUT_FLOATS = KK * (QPB + 1);
    __shared__ float smem[BLK_FLOATS > OUT_FLOATS ? BLK_FLOATS : OUT_FLOATS];
    float* blk = smem + (threadIdx.x >> 6) * BLK_STRIDE;
    float (*outs)[QPB + 1] = reinterpret_cast<float (*)[QPB + 1]>(smem);

    MV_SMALL_KERNEL_PRIO();
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int slice = H2 * W2;
    const int cyl = lane / BS, cxl = lane - cyl * BS;
    const float wm1 = (float)(W2 - 1), hm1 = (float)(H2 - 1);
    const int asq = lane / (2 * K), aa = lane - asq * (2 * K);
    const bool a_is_y = aa >= K;
    const int aoff = (a_is_y ? aa - K : aa) - R;
    const float adim = a_is_y ? hm1 : wm1;

    // ---- the three pipeline stages' state: 0 = being processed, 1 = cells in flight, 2 = coordinates in flight
    auto group_of = [&](int gi, int& b, int& q0) { b = gi / gpb; q0 = (gi - b * gpb) * QPB; };
    // Every load of the pipeline is UNCONDITIONAL (clamped address + select): loads inside per-lane branches make hipcc's wait-count pass
    // give up counting and place `s_waitcnt vmcnt(0)` in front of the current group's LDS writes — which waits for the NEXT group's cells as
    // well and serialises exactly what this kernel exists to overlap (seen in the first ISA dump of this probe).
    auto load_coords = [&](int b, int q0, float& x, float& y) {
        const int qmine = q0 + wave * QPW + (lane & (QPW - 1));
        const int qc = min(qmine, N1 - 1);
        const float xr = coords[((size_t)b * 2 + 0) * N1 + qc];
        const float yr = coords[((size_t)b * 2 + 1) * N1 + qc];
        x = qmine < N1 ? xr : 0.f;
        y = qmine < N1 ? yr : 0.f;
    };
    auto derive = [&](float x, float y, int& bx, int& by, int& margin) {
        const float xc = fminf(fmaxf(x, -1.0e6f), 1.0e6f), yc = fminf(fmaxf(y, -1.0e6f), 1.0e6f);
        bx = ((xc == xc) ? (int)floorf(xc) : 0) - R - 1;
        by = ((yc == yc) ? (int)floorf(yc) : 0) - R - 1;
        const float frx = xc - floorf(xc), fry = yc - floorf(yc);
        margin = ((frx < 0.01f) ? 1 : 0) | ((frx > 0.99f) ? 2 : 0) | ((fry < 0.01f) ? 4 : 0) | ((fry > 0.99f) ? 8 : 0);
    };
    // `live` (uniform): false = the pipeline's tail, every lane reads the zero word — the loads stay unconditional, see above
    auto issue = [&](bool live, int b, int q0, int bx, int by, int margin, float (&v)[QPW][NROUND]) {
#pragma unroll
        for (int s = 0; s < QPW; ++s) {
            const int sbx = __builtin_amdgcn_readlane(bx, s), sby = __builtin_amdgcn_readlane(by, s);
            const int q = q0 + wave * QPW + s;
            const VT* __restrict__ base = vol + ((size_t)b * N1 + q) * slice;
            const int gx = sbx + cxl;
            const int mg = __builtin_amdgcn_readlane(margin, s);
            const bool okx = live && lane < LPR && q < N1 && gx >= 0 && gx < W2 && (cxl != 0 || (mg & 1)) && (cxl != BS - 1 || (mg & 2));
#pragma unroll
            for (int k = 0; k < NINNER; ++k) {
                const int row = 1 + k * RPR + cyl, gy = sby + row;
                const bool ok = okx && gy >= 0 && gy < H2;
                const VT* p = ok ? base + (gy * W2 + gx) : zero;     // masked lanes read a zero word: no select on the loaded value, which
                v[s][k] = (float)*p;                                 // would pull its s_waitcnt in front of the next query's loads
            }
            {                                                        // the margin rows: rare, but issued always (a branch around the load would
                const int row = cyl == 0 ? 0 : BS - 1, gy = sby + row;   // merge its result through a copy, i.e. a wait inside the issue phase)
                const bool ok = okx && (mg & 12) && cyl < 2 && ((mg >> (cyl == 0 ? 2 : 3)) & 1) && gy >= 0 && gy < H2;
                const VT* p = ok ? base + (gy * W2 + gx) : zero;
                v[s][NROUND - 1] = (float)*p;
            }
        }
    };

    int g0 = blockIdx.x;
    if (g0 >= total) return;
    const int G = gridDim.x;
    // stage 0 = being processed, 1 = cells in flight, 2 = coordinates in flight.  Stages 1 / 2 always hold SOME valid group (past the end of
    // the walk: the last real one again), so that every load of the loop body is issued unconditionally.
    int b0, q00, b1, q01;
    float x0, y0, x1, y1, x2, y2;
    int bx0, by0, m0, bx1, by1, m1;
    float v0[QPW][NROUND], v1[QPW][NROUND];
    group_of(g0, b0, q00);
    load_coords(b0, q00, x0, y0);
    derive(x0, y0, bx0, by0, m0);
    issue(true, b0, q00, bx0, by0, m0, v0);
    int g1 = g0 + G;
    bool have1 = g1 < total;                         // (uniform per workgroup)
    int gi1 = have1 ? g1 : g0;
    group_of(gi1, b1, q01);
    load_coords(b1, q01, x1, y1);

    for (;;) {
        const int g2 = g1 + G;
        const bool have2 = have1 && g2 < total;
        const int gi2 = have2 ? g2 : gi1;
        int b2, q02;
        group_of(gi2, b2, q02);
        derive(x1, y1, bx1, by1, m1);                // the NEXT group's cells go out before this group's LDS phases
        issue(have1, b1, q01, bx1, by1, m1, v1);
        load_coords(b2, q02, x2, y2);

        // ================= group 0 of the pipeline: exactly corr_lookup_kernel from its axis pass on =================
        float ax_w[APASS];
        int ax_c[APASS];
#pragma unroll
        for (int ps = 0; ps < APASS; ++ps) {
            const int s = ps * QPA + asq;
            const float qx = __shfl(x0, s, 64), qy = __shfl(y0, s, 64);
            const int obx = __shfl(bx0, s, 64), oby = __shfl(by0, s, 64);
            const float cs = (a_is_y ? qy : qx) + (float)aoff;
            const float g = (2.f * cs) / adim - 1.f;
            const float ic = (g + 1.f) * (adim / 2.f);
            const float f0 = floorf(ic);
            ax_w[ps] = ic - f0;
            const int c = (int)fminf(fmaxf(f0, -2.0e6f), 2.0e6f) - (a_is_y ? oby : obx);
            const bool inb = c >= 0 && c <= BS - 2;
            ax_c[ps] = min(max(c, 0), BS - 2) | (inb ? 256 : 0);
        }
#pragma unroll
        for (int s = 0; s < QPW; ++s) {
#pragma unroll
            for (int k = 0; k < NINNER; ++k)
                if (lane < LPR) blk[s * CELLS + BS + k * LPR + lane] = v0[s][k];
            if (lane < 2 * BS) blk[s * CELLS + (lane < BS ? lane : (BS - 2) * BS + lane)] = v0[s][NROUND - 1];
        }
        __syncthreads();
        float res[TAP_ROUNDS][QPW];
#pragma unroll
        for (int tr = 0; tr < TAP_ROUNDS; ++tr) {
            const int tap = tr * 64 + lane;
            const int ti = tap / K, tj = tap - ti * K;
#pragma unroll
            for (int s = 0; s < QPW; ++s) {
                const int ps = s / QPA, sq = s - ps * QPA;
                const int srcx = sq * 2 * K + ti, srcy = sq * 2 * K + K + tj;
                const float w = __shfl(ax_w[ps], srcx, 64), n = __shfl(ax_w[ps], srcy, 64);
                const int pcx = __shfl(ax_c[ps], srcx, 64), pcy = __shfl(ax_c[ps], srcy, 64);
                const bool inblk = ((pcx & pcy) & 256) != 0;
                const float* p = &blk[s * CELLS + (pcy & 255) * BS + (pcx & 255)];
                const float e = 1.f - w, so = 1.f - n;
                const float vnw = inblk ? p[0] : 0.f, vne = inblk ? p[1] : 0.f;
                const float vsw = inblk ? p[BS] : 0.f, vse = inblk ? p[BS + 1] : 0.f;
                float r0 = vnw * (so * e);
                r0 = r0 + vne * (so * w);
                r0 = r0 + vsw * (n * e);
                r0 = r0 + vse * (n * w);
                res[tr][s] = r0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int tr = 0; tr < TAP_ROUNDS; ++tr) {
            const int tap = tr * 64 + lane;
            if (tap < KK) {
#pragma unroll
                for (int s = 0; s < QPW; ++s) outs[tap][wave * QPW + s] = res[tr][s];
            }
        }
        __syncthreads();
        // ================= rotate the register state BEFORE the token stores: the waits this needs then cover loads only (the stores
        // are younger than every load in flight; behind them a wait on a load would also wait for their write acknowledgements) ==========
        const int sb = b0, sq = q00;
        const bool last = !have1;
        b0 = b1; q00 = q01; x0 = x1; y0 = y1; bx0 = bx1; by0 = by1; m0 = m1;
#pragma unroll
        for (int s = 0; s < QPW; ++s)
#pragma unroll
            for (int k = 0; k < NROUND; ++k) v0[s][k] = v1[s][k];
        g1 = g2; gi1 = gi2; have1 = have2; b1 = b2; q01 = q02; x1 = x2; y1 = y2;
        // (pin the rotated values HERE: hipcc otherwise implements the loop-carried copies on the back edge, behind the stores)
#pragma unroll
        for (int s = 0; s < QPW; ++s)
#pragma unroll
            for (int k = 0; k < NROUND; ++k) asm volatile("" : "+v"(v0[s][k]));
        asm volatile("" : "+v"(x1), "+v"(y1));
        for (int idx = t; idx < KK * QPB; idx += NTHR) {
            const int k = idx / QPB, c = idx - k * QPB;
            if (sq + c < N1) out[((size_t)sb * KK + k) * N1 + sq + c] = outs[k][c];
        }
        if (last) break;
        __syncthreads();                             // the transposed outputs have been read: the region becomes the staging buffer again
    }
}

}  // namespace

// variant 0: mv_corr_lookup's batched kernel <4, 4, 32>; 1: persistent <4, 4, 32>, `wgs_per_cu` workgroups per CU; 2: persistent <4, 8, 64>
extern "C" int probe_lookup_launch(const float* vol, const float* coords, float* out, int B, int N1, int H2, int W2, int variant, int wgs_per_cu,
                                   const float* zero /* device: one float 0 */, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    if (variant == 0) {
        hipLaunchKernelGGL((corr_lookup_kernel<4, 4, 32>), dim3((N1 + 31) / 32, B), dim3(512), 0, s, vol, coords, out, N1, H2, W2);
    } else if (variant >= 10) {       // other tilings of the SHIPPED kernel (more queries in flight per CU: fewer threads per query)
#define LT(QPW, QPB) hipLaunchKernelGGL((corr_lookup_kernel<4, QPW, QPB>), dim3((N1 + QPB - 1) / QPB, B), dim3(64 * (QPB / QPW)), 0, s, vol, coords, out, N1, H2, W2)
        switch (variant) {
            case 10: LT(8, 32); break;     // 256 threads, 8 queries per wave
            case 11: LT(8, 64); break;     // 512 threads
            case 12: LT(4, 16); break;     // 256 threads, 16-query groups
            case 13: LT(16, 64); break;    // 256 threads, 16 queries per wave
            case 14: LT(8, 16); break;     // 128 threads
            default: LT(2, 16); break;     // 15: the one-frame variant at this batch size
        }
#undef LT
    } else if (variant == 1) {
        const int gpb = (N1 + 31) / 32, total = B * gpb;
        const int grid = total < cus * wgs_per_cu ? total : cus * wgs_per_cu;
        hipLaunchKernelGGL((corr_lookup_persist_kernel<4, 4, 32>), dim3(grid), dim3(512), 0, s, vol, coords, out, N1, H2, W2, gpb, total, zero);
    } else {
        const int gpb = (N1 + 63) / 64, total = B * gpb;
        const int grid = total < cus * wgs_per_cu ? total : cus * wgs_per_cu;
        hipLaunchKernelGGL((corr_lookup_persist_kernel<4, 8, 64>), dim3(grid), dim3(512), 0, s, vol, coords, out, N1, H2, W2, gpb, total, zero);
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
