// Hand-counted memory waits, LDS-DMA and the other inline-asm pieces the volume kernels share (corr_volume.hip,
// corr_volume_split.hip).  gfx950 only.
#pragma once
#include "common.h"

void mv_note_volume_kernel(const char* name);   // corr_volume.hip: what mv_corr_volume_last_kernel reports

// ---- hand-counted memory waits and LDS-DMA (used by the DMA-staged tile below and by the streaming kernels) ----
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait for this wave's older memory operations (all but the newest N), then the workgroup barrier — one statement so that
// nothing can be scheduled between the two
template <int N>
__device__ __forceinline__ void wait_vmcnt_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
// LDS-DMA: 64 lanes x 16 B from (uniform base + per-lane 32-bit offset) to LDS bytes [lds_dst, lds_dst + 1024) in lane order.  M0
// carries the destination and is compiler-reserved: saved, written and restored inside the one statement.  Invisible to
// hipcc's s_waitcnt bookkeeping (counted by hand at the call sites).
__device__ __forceinline__ void glds16_s(unsigned voff, const void* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

// The same copy as a GROUP of up to four 1-KB pieces that share one M0 value and one scalar base: the instruction's immediate offset is
// added to BOTH the global and the LDS address, so pieces at (src + i KB -> dst + i KB), i = 0..3, need no SALU of their own.
// glds16_m0 writes M0 and LEAVES it (no save / restore: the kernels that use this form contain no other M0 user — checked in the ISA
// dump: the only `m0` writes of corr_volume_split_stream are these); glds16_next relies on that M0.
__device__ __forceinline__ void glds16_m0(unsigned voff, const void* sbase, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int OFF>
__device__ __forceinline__ void glds16_next(unsigned voff, const void* sbase) {
    static_assert(OFF > 0 && OFF < 4096, "13-bit signed immediate");
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(voff), "s"(sbase), "n"(OFF) : "memory");
}

// LDS-DMA: 64 lanes x 16 B from per-lane global addresses to LDS bytes [lds_dst, lds_dst + 1024) in lane order.  M0 carries the
// destination and is compiler-reserved: saved, written and restored inside the one statement.  Invisible to hipcc's s_waitcnt
// bookkeeping (counted by hand, see the kernel).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

