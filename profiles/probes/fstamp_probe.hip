#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <type_traits>
#define MV_VOL_STORE_ASM_MOD " nt"
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void wait_vmcnt_barrier() { asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory"); }
__device__ __forceinline__ void glds16_s(unsigned voff, const void* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

template <int C>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void corr_volume_f32_stream(
    const float* __restrict__ f1, const float* __restrict__ f2, float* __restrict__ out, int N, int B, int R, long long* __restrict__ ts) {
    int tsn = 0;
    const bool rec = (threadIdx.x == 0) && (blockIdx.x % 37 == 0);
    long long* tsp = ts + (blockIdx.x / 37) * 256;
#define STAMP() do { if (rec && tsn < 250) tsp[tsn++] = clock64(); } while (0)
    STAMP();
    constexpr int KP = C / 2, KPH = KP / 2;      // k pairs; per ring slot
    constexpr int SLOTF = (C / 2) * 64;          // floats per slot: C / 2 k rows x 64 columns
    constexpr int NPC = SLOTF * 4 / 1024 / 4;    // 1-KB DMA pieces per wave and slot (8)
    static_assert(KPH > NPC + 1, "interleave plan: pieces behind k pairs 0..NPC-1, loader walk behind NPC");
    extern __shared__ __attribute__((aligned(16))) float smem_fs[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int kh = lane >> 5, li = lane & 31;
    const int nb = (N + 127) >> 7, nc = N >> 6;
    const int per = nb * nc, T = B * per;
    int it, it_end;
    {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3, nj = gridDim.x >> 3;
        const long lo = (long)x * T / 8, hi = ((long)x + 1) * T / 8;
        it = (int)(lo + (hi - lo) * j / nj);
        it_end = (int)(lo + (hi - lo) * (j + 1) / nj);
    }
    if (it >= it_end) return;
    auto reg_c0 = [&](int g) { return (int)((long)g * nc / R); };
    auto decode = [&](int i, int& b, int& g, int& band, int& c) {
        b = i / per;
        int rem = i - b * per;
        g = 0;
        while (g + 1 < R && rem >= nb * reg_c0(g + 1)) ++g;
        rem -= nb * reg_c0(g);
        const int w = reg_c0(g + 1) - reg_c0(g);
        band = rem / w;
        c = reg_c0(g) + (rem - band * w);
        b = __builtin_amdgcn_readfirstlane(b);
        g = __builtin_amdgcn_readfirstlane(g);
        band = __builtin_amdgcn_readfirstlane(band);
        c = __builtin_amdgcn_readfirstlane(c);
    };
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem_fs);
    // ---- B loader (LDS-DMA): walks (item, K half) one step ahead of the MFMAs, all state wave-uniform ----
    int ld_it = it, ld_b, ld_g, ld_band, ld_c, ld_c0, ld_cend, ld_h = 0;
    decode(it, ld_b, ld_g, ld_band, ld_c);
    ld_c0 = reg_c0(ld_g);
    ld_cend = reg_c0(ld_g + 1);
    bool ld_live = true;                         // false once the run's last half has been issued
    // a piece = 4 k rows x 64 columns: lane -> row lane / 16, columns 4 (lane % 16) ..
    const unsigned pc_vo = (unsigned)(((lane >> 4) * N + 4 * (lane & 15)) * 4);
    const float* pc_src = nullptr;               // per half: source of this wave's piece 0 / LDS byte address of it
    unsigned pc_dst = 0;
    auto arm = [&]() __attribute__((always_inline)) {
        pc_src = f2 + ((size_t)ld_b * C + ld_h * (C / 2) + 4 * wave * NPC) * N + (size_t)ld_c * 64;
        pc_dst = lds0 + (unsigned)(ld_h * SLOTF + wave * NPC * 256) * 4u;
    };
    auto piece = [&](int p) __attribute__((always_inline)) {
        if (ld_live) glds16_s(pc_vo, pc_src + (size_t)(4 * p) * N, pc_dst + (unsigned)p * 1024u);
    };
    auto advance = [&]() __attribute__((always_inline)) {           // behind the last piece of a half
        if (ld_h == 0) {
            ld_h = 1;
        } else {
            ld_h = 0;
            if (ld_it + 1 >= it_end) {
                ld_live = false;
            } else {
                ++ld_it;
                if (++ld_c == ld_cend) {
                    if (++ld_band == nb) {
                        ld_band = 0;
                        if (++ld_g == R) {
                            ld_g = 0;
                            ++ld_b;
                        }
                        ld_c0 = reg_c0(ld_g);
                        ld_cend = reg_c0(ld_g + 1);
                    }
                    ld_c = ld_c0;
                }
            }
        }
        arm();
    };
    // ---- A fragments: ordinary (compiler-tracked) loads.  An asm load's destination counts as written at the end of the
    // statement, and under register pressure hipcc parks such values elsewhere right away — before the data has landed (seen with
    // an earlier form: the first items after a band change came out with the previous band's rows). ----
    float af[KP];
    auto load_a = [&](int b, int band) __attribute__((always_inline)) {
        const int row = min(band * 128 + wave * 32 + li, N - 1);
        // buffer loads: descriptor + one 32-bit lane offset + a scalar offset per k pair (a global_load would carry a 64-bit
        // per-lane address for each of the C / 2 loads: 2 x 128 VGPRs of addresses and a spilled fragment array)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(f1 + (size_t)b * C * N), 0, C * N * 4, 0x00020000);
        const int vo = (kh * N + row) * 4;
#pragma clang loop unroll(full)
        for (int ks = 0; ks < KP; ++ks) af[ks] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, 2 * ks * N * 4, 0));
    };
    unsigned roff[16];                           // per-lane byte offsets of the 16 accumulator rows (C/D layout), fixed for a band
    float* O = nullptr;                          // column 0 of the current item's block, row 0 of the pair
    f32x16 c0, c1;
    // one K half: 2 x KPH MFMAs; the other ring slot is refilled behind the first ones
    auto half = [&](auto HH) __attribute__((always_inline)) {
        constexpr int H = decltype(HH)::value;
        // H = 0: this wave's pieces of slot 0 went out before the previous item's 32 stores; H = 1: nothing was issued behind the
        // pieces of slot 1.  Behind the barrier all four waves' pieces are in and the other slot is free.
        STAMP();
        wait_vmcnt_barrier<(H == 0 ? 32 : 0)>();
        STAMP();
        const float* q = smem_fs + H * SLOTF + kh * 64 + li;
        constexpr int PF = 2;
        float fb[PF + 1][2];
#pragma unroll
        for (int ks = 0; ks < PF; ++ks) {
            fb[ks][0] = q[ks * 128];
            fb[ks][1] = q[ks * 128 + 32];
        }
#pragma clang loop unroll(full)
        for (int ks = 0; ks < KPH; ++ks) {
            const float a = af[H * KPH + ks];
            if (H == 0 && ks == 0) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, fb[0][0], z, 0, 0, 0);
            } else {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, fb[ks % (PF + 1)][0], c0, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ks < NPC) piece(ks);
            if (ks == NPC) advance();
            __builtin_amdgcn_sched_barrier(0);
            if (H == 0 && ks == 0) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, fb[0][1], z, 0, 0, 0);
            } else {
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, fb[ks % (PF + 1)][1], c1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ks + PF < KPH) {
                fb[(ks + PF) % (PF + 1)][0] = q[(ks + PF) * 128];
                fb[(ks + PF) % (PF + 1)][1] = q[(ks + PF) * 128 + 32];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    int cur_key = -1;
    arm();
#pragma unroll
    for (int p = 0; p < NPC; ++p) piece(p);      // prologue: first half of the first item into slot 0
    advance();
    for (; it < it_end; ++it) {
        int b, g, band, c;
        decode(it, b, g, band, c);
        const int key = b * nb + band;
        if (key != cur_key) {                    // band change: A fragments, store rows
            load_a(b, band);
#pragma clang loop unroll(full)
            for (int ks = 0; ks < KP; ++ks) asm volatile("" : "+v"(af[ks]));   // hipcc's wait for them lands here, inside the branch
            cur_key = key;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                roff[r] = ((unsigned)min(band * 128 + wave * 32 + 4 * kh + (r & 3) + 8 * (r >> 2), N - 1) * (unsigned)N + li) * 4u;
        }
        O = out + (size_t)b * N * N + (size_t)c * 64;
        STAMP();
        half(H0{});
        half(H1{});
        STAMP();
        // The asm stores below read the accumulators straight behind the last MFMAs.  hipcc's hazard recognizer does not look
        // into inline asm, and "XDL write VGPR -> VMEM read of it" is a software hazard on CDNA (18 wait states for a 16-pass
        // MFMA): without the s_nops the last MFMA's block was stored before it had been written (seen: 0.3 % wrong elements,
        // all in the second column block).
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r) {           // asm: uniform base + 32-bit lane offset (see the 16-bit kernel); counted by hand
            asm volatile("global_store_dword %0, %1, %2" MV_VOL_STORE_ASM_MOD ::"v"(roff[r]), "v"(c0[r]), "s"(O) : "memory");
            asm volatile("global_store_dword %0, %1, %2 offset:128" MV_VOL_STORE_ASM_MOD ::"v"(roff[r]), "v"(c1[r]), "s"(O) : "memory");
        }
    }
    STAMP();
    if (rec) tsp[255] = tsn;
    wait_vmcnt<0>();                             // nothing of this workgroup may still be in flight towards its LDS when it retires
}


extern "C" int run(const float* a, const float* b, float* out, int N, int B, int R, int grid, hipStream_t s, long long* ts) {
    static bool done = false; if (!done) { hipFuncSetAttribute((const void*)corr_volume_f32_stream<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); done = true; }
    hipLaunchKernelGGL((corr_volume_f32_stream<256>), dim3(grid), dim3(256), 64 * 1024, s, a, b, out, N, B, R, ts);
    return (int)hipGetLastError();
}
