// (f)2 — the fused cost patch embedding for 60 / 64 x 80 slices, PIPELINED across slices by two wave groups (round 5).
//
// patch_embed.hip runs staging -> conv1 -> conv2 -> conv3 as PHASES of one 4-wave workgroup, barrier to barrier.  Its counters after the bank-conflict work
// (profiles/r05_pmc_patch_embed.json): LDS-array cycles halved (1.27e8 -> 6.3e7 per launch), matrix pipe busy 36-38 %, duration UNCHANGED (~330 us): the kernel is
// bound by the phase structure — conv1 is VALU / LDS-issue work (640 MFMA cycles per slice under ~4 k cycles of address arithmetic, 16-bit conversions and stores),
// conv2 and conv3 are matrix work, and at any time all four waves are in the same phase: the matrix pipe idles through conv1, the VALU through conv3.
//
// Here one 8-wave workgroup per CU (two waves per SIMD) runs the layers of DIFFERENT slices at the same time:
//
//   step k:   FRONT group (waves 0-3): slice k      conv1 -> conv1 map      | slice k + 1 -> input window; conv2 -> conv2 map [k & 1]
//             BACK  group (waves 4-7): slice k - 1  conv3 taps 0 .. 11      | conv3 taps 12 .. 35, tokens -> HBM                 (from conv2 map [(k - 1) & 1])
//                                                                     barrier                                              barrier
//
//   so a SIMD always holds one wave of VALU / LDS-heavy work next to one wave of MFMA-heavy work.  Same LDS plan as patch_embed.hip (pe::PE: the two conv2
//   maps that held the two slices of a pass are now the ping-pong between the groups).  gfx950 has one workgroup barrier: the back group executes the front group's
//   inner barrier at a fixed point of its tap loop.
//   BACK: wave = one 16-channel tile, all five 16-token tiles (no K split, nothing to exchange: there is no free LDS for it; each activation fragment is read by
//   four waves — the LDS-cycle model (profiles/probes/r5_pe_v2_index_model.py) puts the slice at ~8.3 k LDS-array cycles inside a ~10 k-cycle step).
//   FRONT: conv1 / conv2 of patch_embed.hip with the SWAPPED operand order of patch_embed_v2.hip: a lane ends with four consecutive channels of one pixel = one
//   8-byte ds_write_b64 per tile instead of four ds_write_b16.
// Measured (profiles/r05_patch_embed_pipeline_ab.log; 9600 slices, bf16, 16-bit cells / tokens): 295 us (phase kernel) -> 244 us.  Phase knock-outs of this kernel:
// without conv1 204, without conv2 170, without conv3's taps 189, without staging 230, none of them 40 us — the layers' costs still ADD (37 + 71 + 52 + 11 over a
// 40-us skeleton): two waves per SIMD share one matrix pipe (the three layers are 113 us of pure MFMA time at the 2.1 GHz the chip holds here) and one LDS (~8.3 k
// array cycles per slice with the back group's four-fold fragment reads), so the overlap buys latency hiding, not a second resource.
// Results: the same arithmetic as patch_embed.hip (16-bit operands, fp32 accumulation in the same k order per output) — bit-identical tokens (tested).
#include "patch_embed_dev.h"
#include <algorithm>
#include <atomic>
#include <type_traits>
#include <stdlib.h>

using namespace pe;

namespace {

template <int H2, int W2, bool TOKENS, bool F16, bool IN16, bool OUT16>
__global__ __launch_bounds__(512) void cost_patch_embed_pipelined_kernel(const void* __restrict__ vol_, const char* __restrict__ wp, void* __restrict__ out_, int S) {
    using P = PE<H2, W2>;
    extern __shared__ __attribute__((aligned(16))) char smem_pe3[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grp = wave >> 2, w4 = wave & 3, tg = t & 255;
    const int n16 = lane & 15, g4 = lane >> 4;
    char* const in0 = smem_pe3 + P::OFF_IN0;
    char* const o1 = smem_pe3 + P::OFF_O1;
    char* const o2 = smem_pe3 + P::OFF_O2;
    const char* const vol = reinterpret_cast<const char*>(vol_);
    constexpr int EPL = IN16 ? 8 : 4;
    constexpr int QN = H2 * W2 / EPL, NPRE = (QN + 255) / 256;
    static_assert(W2 % EPL == 0 && NPRE <= 5, "slice staging");

    // ---- once per workgroup: zero the activation buffers (their halos stay zero), conv2's second channel tile -> LDS
    for (unsigned a = (unsigned)t * 16u; a < P::OFF_W2B; a += 512u * 16u) *reinterpret_cast<i32x4*>(smem_pe3 + a) = i32x4{0, 0, 0, 0};
    for (unsigned a = (unsigned)t * 16u; a < P::W2B_BYTES; a += 512u * 16u)
        *reinterpret_cast<i32x4*>(smem_pe3 + P::OFF_W2B + a) = *reinterpret_cast<const i32x4*>(wp + PE_W2_OFF + 18 * 1024 + a);
    const float* bias = reinterpret_cast<const float*>(wp + PE_B_OFF);
    __syncthreads();

    // slices of this workgroup: blockIdx.x, + gridDim.x, ...; step k: front = local slice k, back = local slice k - 1
    const int n_local = (S - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

    if (grp == 0) {
        // =============================================================== FRONT: staging, conv1, conv2
        i32x4 w2f[18];
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) w2f[ks] = *reinterpret_cast<const i32x4*>(wp + PE_W2_OFF + ((size_t)ks * 64 + lane) * 16);
        i32x4 w1f[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) w1f[s] = *reinterpret_cast<const i32x4*>(wp + PE_W1_OFF + (s * 64 + lane) * 16);
        float b1v[4], b2v[2][4];                                              // swapped products: a lane owns channels 4 g4 + e of its tile
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            b1v[e] = bias[4 * g4 + e];
            b2v[0][e] = bias[32 + 4 * g4 + e];
            b2v[1][e] = bias[48 + 4 * g4 + e];
        }
        // conv1 addressing: tile j = 5 m + r of this wave = pixels (w4 + 4 r + 20 m) 16 ..: 20 tiles = 8 output rows, so tile 5 m + r = tile r + m constant strides
        static_assert(P::T1 % 20 == 0 && (20 * 16) % P::W1 == 0, "conv1: five tiles per wave span whole output rows");
        int c1_a[5], c1_d[5];
#pragma unroll
        for (int r5 = 0; r5 < 5; ++r5) {
            const int p = (w4 + 4 * r5) * 16 + n16, oy = p / P::W1, ox = p - oy * P::W1;
            c1_a[r5] = ((2 * oy + 4 * (g4 & 1) + (g4 >> 1)) * P::IN_PITCH + 2 * ox) * 2;              // ky of k-step 0 (mv_patch_embed_pack); k-step 1: + 2 rows
            c1_d[r5] = (int)P::o1_cell(g4 >> 1, oy + 2, ox + 2) + (g4 & 1) * 8;                        // channels 4 g4 .. + 3 of pixel (oy, ox)
        }
        i32x4 pre[NPRE];
        auto fetch = [&](int k) {
            const int s = (int)blockIdx.x + k * (int)gridDim.x;
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const int q = tg + 256 * i;
                pre[i] = (k < n_local && q < QN) ? __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(vol + (size_t)s * (H2 * W2) * (IN16 ? 2 : 4)) + q)
                                                  : i32x4{0, 0, 0, 0};
            }
        };
        auto stage = [&]() __attribute__((always_inline)) {                  // registers -> in0 (16-bit)
#pragma unroll
                for (int i = 0; i < NPRE; ++i) {
                    const int q = tg + 256 * i;
                    if (q < QN) {
                        const int y = q / (W2 / EPL), x = EPL * (q - y * (W2 / EPL));
                        unsigned* d = reinterpret_cast<unsigned*>(in0 + ((y + 2) * P::IN_PITCH + x + 2) * 2);
                        if constexpr (IN16) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) d[e] = (unsigned)pre[i][e];
                        } else {
                            const f32x4 v = __builtin_bit_cast(f32x4, pre[i]);
                            d[0] = cvt_pack<F16>(v[0], v[1]);
                            d[1] = cvt_pack<F16>(v[2], v[3]);
                        }
                    }
                }
        };
        fetch(0);
        if (n_local > 0) { stage(); fetch(1); }
        __syncthreads();                                                     // B0: slice 0 staged
        for (int k = 0; k <= n_local; ++k) {
            const bool live = k < n_local;                                   // (the last step only drains the back group)
            int gv = g4;
            asm volatile("" : "+v"(gv));                                     // opaque per step: keeps ~100 epilogue addresses from being hoisted (spills)
            // ---- (B) conv1: 20 tiles per wave, 2 k-steps.  Software-pipelined by hand: the fragment of tile j + 1 is read BEFORE tile j's store — hipcc keeps
            // LDS reads behind earlier LDS writes it cannot disambiguate, which serialised the 20 tiles (read -> wait -> 2 MFMAs -> convert -> store, one LDS round
            // trip each: ~6 k cycles per slice for 640 cycles of MFMA)
            if (live) {
                constexpr int NT = P::T1 / 4;                               // 20 tiles: j = 5 m + r5
                auto load_tile = [&](int j, i32x4* af) __attribute__((always_inline)) {
                    const char* a0 = in0 + c1_a[j % 5] + (j / 5) * (16 * P::IN_PITCH * 2);
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const unsigned* ap = reinterpret_cast<const unsigned*>(a0 + 2 * s * P::IN_PITCH * 2);
                        af[s] = i32x4{(int)ap[0], (int)ap[1], (int)ap[2], (int)ap[3]};
                    }
                };
                i32x4 afr[3][2];
                load_tile(0, afr[0]);
                load_tile(1, afr[1]);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (j + 2 < NT) load_tile(j + 2, afr[(j + 2) % 3]);
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 2; ++s) acc = mma16t<F16>(__builtin_bit_cast(bf16x8, afr[j % 3][s]), __builtin_bit_cast(bf16x8, w1f[s]), acc);
                    const unsigned lo = cvt_pack<F16>(fmaxf(acc[0] + b1v[0], 0.f), fmaxf(acc[1] + b1v[1], 0.f));
                    const unsigned hi = cvt_pack<F16>(fmaxf(acc[2] + b1v[2], 0.f), fmaxf(acc[3] + b1v[3], 0.f));
                    *reinterpret_cast<unsigned long long*>(o1 + c1_d[j % 5] + (j / 5) * (8 * P::O1_XH * 16)) = (unsigned long long)lo | ((unsigned long long)hi << 32);
                }
            }
            __syncthreads();                                                 // B2
            // ---- (A') the input window is free (conv1 of slice k is done): slice k + 1 -> in0 now, beside the back group's taps, slice k + 2 -> registers
            if (k + 1 < n_local) { stage(); fetch(k + 2); }
            // ---- (C) conv2: this wave = output rows 4 w4 .. 4 w4 + 3 (five 16-pixel tiles) x both 16-channel tiles; k-step = two neighbouring taps x 16 channels
            if (live) {
                f32x4 acc[5][2];
                const char* abase[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const int p = (5 * w4 + i) * 16 + n16, oy = p / P::W2o, ox = p - oy * P::W2o;
                    abase[i] = o1 + P::o1_cell(g4 & 1, 2 * oy, 2 * ox) + (g4 >> 1) * P::O1_PLANE;
                }
                constexpr int PFA = 1;
                bf16x8 af[PFA + 1][5], bf1[PFA + 1];
                const char* const wb1 = smem_pe3 + P::OFF_W2B + lane * 16;
                auto fetch_k = [&](int ks) __attribute__((always_inline)) {
                    bf1[ks % (PFA + 1)] = *reinterpret_cast<const bf16x8*>(wb1 + ks * 1024);
#pragma unroll
                    for (int i = 0; i < 5; ++i) af[ks % (PFA + 1)][i] = *reinterpret_cast<const bf16x8*>(abase[i] + ((ks / 3) * P::O1_XH + ks % 3) * 16);
                };
#pragma unroll
                for (int ks = 0; ks < PFA; ++ks) fetch_k(ks);
#pragma unroll
                for (int ks = 0; ks < 18; ++ks) {
                    if (ks + PFA < 18) fetch_k(ks + PFA);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        acc[i][0] = mma16t<F16>(af[ks % (PFA + 1)][i], __builtin_bit_cast(bf16x8, w2f[ks]), acc[i][0]);
                        acc[i][1] = mma16t<F16>(af[ks % (PFA + 1)][i], bf1[ks % (PFA + 1)], acc[i][1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                char* const o2k = o2 + (k & 1) * P::O2_BYTES + (gv >> 1) * 2 * P::O2_PLANE + (gv & 1) * 8;
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const int pp = (5 * w4 + i) * 16 + n16, y = pp / P::W2o, x = pp - y * P::W2o;
                    char* d = o2k + P::o2_cell(0, y + 2, x + 2);
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const unsigned lo = cvt_pack<F16>(fmaxf(acc[i][nt][0] + b2v[nt][0], 0.f), fmaxf(acc[i][nt][1] + b2v[nt][1], 0.f));
                        const unsigned hi = cvt_pack<F16>(fmaxf(acc[i][nt][2] + b2v[nt][2], 0.f), fmaxf(acc[i][nt][3] + b2v[nt][3], 0.f));
                        *reinterpret_cast<unsigned long long*>(d + nt * 4 * P::O2_PLANE) = (unsigned long long)lo | ((unsigned long long)hi << 32);
                    }
                }
            }
            __syncthreads();                                                 // B3: conv2 map [k & 1] complete; the back group is done with map [(k - 1) & 1]
        }
    } else {
        // =============================================================== BACK: conv3 of the previous step's slice; wave = channel tile w4, all five token tiles
        const char* const w3 = wp + PE_W3_OFF + ((size_t)w4 * 36 * 64 + lane) * 16;          // tap at + tap KB
        float b3s[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) b3s[e] = bias[64 + w4 * 16 + 4 * g4 + e];
        const float b3n = bias[64 + w4 * 16 + n16];
        const char* abase0[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int q = i * 16 + n16, oy = q / P::W3, ox = q - oy * P::W3;
            abase0[i] = o2 + P::o2_cell(g4, 2 * oy, 2 * ox);
        }
        __syncthreads();                                                     // B0
        for (int k = 0; k <= n_local; ++k) {
            const bool live = k >= 1;
            int gv = g4;
            asm volatile("" : "+v"(gv));
            const unsigned mapoff = ((k - 1) & 1) * P::O2_BYTES;
            f32x4 acc[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            constexpr int PF = 8, PFA = 1;
            i32x4 bq[PF];
            bf16x8 af[PFA + 1][5];
            auto fetch_a = [&](int tap) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 5; ++i) af[tap % (PFA + 1)][i] = *reinterpret_cast<const bf16x8*>(abase0[i] + mapoff + P::o2_cell(0, tap / 6, tap % 6));
            };
            auto taps = [&](auto t0c, auto t1c) __attribute__((always_inline)) {     // (compile-time bounds: bq / af are indexed by constants)
                constexpr int t0 = decltype(t0c)::value, t1 = decltype(t1c)::value;
#pragma unroll
                for (int kk = t0; kk < t1; ++kk) {
                    const i32x4 b0 = bq[kk % PF];
                    if (kk + PFA < 36) fetch_a(kk + PFA);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        if constexpr (TOKENS) acc[i] = mma16t<F16>(af[kk % (PFA + 1)][i], __builtin_bit_cast(bf16x8, b0), acc[i]);
                        else acc[i] = mma16<F16>(af[kk % (PFA + 1)][i], __builtin_bit_cast(bf16x8, b0), acc[i]);
                    }
                    if (kk + PF < 36) bq[kk % PF] = *reinterpret_cast<const i32x4*>(w3 + (size_t)(kk + PF) * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (live) {
#pragma unroll
                for (int kk = 0; kk < PF; ++kk) bq[kk] = *reinterpret_cast<const i32x4*>(w3 + (size_t)kk * 1024);
            }
            if (live) {
#pragma unroll
                for (int kk = 0; kk < PFA; ++kk) fetch_a(kk);
                taps(std::integral_constant<int, 0>{}, std::integral_constant<int, 12>{});
            }
            __syncthreads();                                                 // B2 (conv1 of slice k done)
            if (live) {
                taps(std::integral_constant<int, 12>{}, std::integral_constant<int, 36>{});
                const int s_out = (int)blockIdx.x + (k - 1) * (int)gridDim.x;
                const size_t base = (size_t)s_out * (P::M3 * 64);
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    if constexpr (TOKENS) {
                        // lane: token i * 16 + n16, channels w4 * 16 + 4 g4 + e
                        const size_t at = base + (size_t)(i * 16 + n16) * 64 + w4 * 16 + 4 * gv;
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][e] + b3s[e];
                        if constexpr (OUT16) {
                            unsigned* d = reinterpret_cast<unsigned*>(reinterpret_cast<uint16_t*>(out_) + at);
                            d[0] = cvt_pack<F16>(v[0], v[1]);
                            d[1] = cvt_pack<F16>(v[2], v[3]);
                        } else {
                            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out_) + at) = v;
                        }
                    } else {
                        // lane: channel w4 * 16 + n16, tokens i * 16 + 4 g4 + e (four consecutive: one 16- / 8-byte store)
                        const size_t at = base + (size_t)(w4 * 16 + n16) * P::M3 + i * 16 + 4 * gv;
                        if constexpr (OUT16) {
                            unsigned* d = reinterpret_cast<unsigned*>(reinterpret_cast<uint16_t*>(out_) + at);
                            d[0] = cvt_pack<F16>(acc[i][0] + b3n, acc[i][1] + b3n);
                            d[1] = cvt_pack<F16>(acc[i][2] + b3n, acc[i][3] + b3n);
                        } else {
                            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out_) + at) = f32x4{acc[i][0] + b3n, acc[i][1] + b3n, acc[i][2] + b3n, acc[i][3] + b3n};
                        }
                    }
                }
            }
            __syncthreads();                                                 // B3
        }
    }
}

template <int H2, bool F16, bool IN16, bool OUT16>
int launch_pipelined(const void* cost_maps, const void* packed, void* out, int S, int token_layout, hipStream_t stream) {
    using P = PE<H2, 80>;
    static std::atomic<bool> attr_done[64];
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[dev].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute((const void*)cost_patch_embed_pipelined_kernel<H2, 80, true, F16, IN16, OUT16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)cost_patch_embed_pipelined_kernel<H2, 80, false, F16, IN16, OUT16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done[dev].store(true, std::memory_order_release);
    }
    int ncu = cus[dev].load(std::memory_order_relaxed);
    if (!ncu) {
        hipDeviceProp_t prop;
        ncu = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        cus[dev].store(ncu, std::memory_order_relaxed);
    }
    const dim3 grid(std::min(S, ncu));                                       // persistent: one 8-wave workgroup per CU
    if (token_layout)
        hipLaunchKernelGGL((cost_patch_embed_pipelined_kernel<H2, 80, true, F16, IN16, OUT16>), grid, dim3(512), P::LDS_BYTES, stream, cost_maps, (const char*)packed, out, S);
    else
        hipLaunchKernelGGL((cost_patch_embed_pipelined_kernel<H2, 80, false, F16, IN16, OUT16>), grid, dim3(512), P::LDS_BYTES, stream, cost_maps, (const char*)packed, out, S);
    return mv_launch_status();
}

template <int H2>
int dispatch_pipelined(const void* cost_maps, int in16, const void* packed, void* out, int out16, int S, int token_layout, int f16, hipStream_t st) {
    if (f16) {
        if (!in16) return launch_pipelined<H2, true, false, false>(cost_maps, packed, out, S, token_layout, st);
        return out16 ? launch_pipelined<H2, true, true, true>(cost_maps, packed, out, S, token_layout, st)
                     : launch_pipelined<H2, true, true, false>(cost_maps, packed, out, S, token_layout, st);
    }
    if (!in16) return launch_pipelined<H2, false, false, false>(cost_maps, packed, out, S, token_layout, st);
    return out16 ? launch_pipelined<H2, false, true, true>(cost_maps, packed, out, S, token_layout, st)
                 : launch_pipelined<H2, false, true, false>(cost_maps, packed, out, S, token_layout, st);
}

}  // namespace

// library-internal (patch_embed.hip dispatches here)
int mv_cost_patch_embed_pipelined(const void* cost_maps, int in16, const void* packed, void* out, int out16, int S, int H2, int W2, int token_layout, int f16,
                                  mvStream_t stream) {
    if (W2 != 80) return MV_ERR_UNSUPPORTED;
    if (H2 == 60) return dispatch_pipelined<60>(cost_maps, in16, packed, out, out16, S, token_layout, f16, (hipStream_t)stream);
    if (H2 == 64) return dispatch_pipelined<64>(cost_maps, in16, packed, out, out16, S, token_layout, f16, (hipStream_t)stream);
    return MV_ERR_UNSUPPORTED;
}
