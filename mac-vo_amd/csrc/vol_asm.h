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

