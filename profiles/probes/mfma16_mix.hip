// scratch: what slows a one-wave-per-SIMD v_mfma_f32_32x32x16_bf16 stream shaped like corr_volume_split_stream's half
// (8 k-steps x 12 MFMAs on 2 accumulators): B operands from LDS (6 ds_read_b128 / k-step), stores from AGPRs (2 / k-step),
// scalar filler.  FLAGS: 1 = B from LDS, 2 = stores, 4 = s_nop filler after each MFMA pair, 8 = 4 accumulators (2 per column block)
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int FLAGS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void mix(const i32x4* in, float* out, int iters, long long* cyc) {
    extern __shared__ i32x4 smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 3072; i += 256) smem[i] = in[i & 1023];
    __syncthreads();
    i32x4 av[48];
    for (int i = 0; i < 48; ++i) { av[i] = in[(threadIdx.x + 64 * i) & 1023]; asm volatile("" : "+a"(av[i])); }
    f32x16 c0, c1, d0, d1, p0, p1;
    for (int q = 0; q < 16; ++q) { c0[q] = c1[q] = d0[q] = d1[q] = 0.f; p0[q] = (float)q; p1[q] = (float)-q; }
    unsigned roff[16];
    for (int r = 0; r < 16; ++r) roff[r] = (unsigned)((blockIdx.x * 256 + threadIdx.x) * 64 + r * 4) * 4u;
    const i32x4* q = smem + lane;
    constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            i32x4 fb[2][2][3];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) fb[0][j][p] = q[((j * 8 + 0) * 3 + p) * 64];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                if (ks + 1 < 8) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int p = 0; p < 3; ++p) fb[nxt][j][p] = (FLAGS & 1) ? q[((j * 8 + ks + 1) * 3 + p) * 64] : fb[cur][j][p];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int qd = 0; qd < 6; ++qd) {
                    const i32x4 a = av[(h * 8 + ks) * 3 + PA[qd]];
                    if ((FLAGS & 8) && (qd & 1)) {
                        d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, fb[cur][0][PB[qd]]), d0, 0, 0, 0);
                        d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, fb[cur][1][PB[qd]]), d1, 0, 0, 0);
                    } else {
                        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, fb[cur][0][PB[qd]]), c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, fb[cur][1][PB[qd]]), c1, 0, 0, 0);
                    }
                    if (FLAGS & 4) asm volatile("s_nop 0\n\ts_nop 0" ::: "memory");
                    if ((FLAGS & 2) && qd == 5) {
                        asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(roff[h * 8 + ks]), "a"(p0[h * 8 + ks]), "s"(out) : "memory");
                        asm volatile("global_store_dword %0, %1, %2 offset:128 nt" ::"v"(roff[h * 8 + ks]), "a"(p1[h * 8 + ks]), "s"(out) : "memory");
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int q2 = 0; q2 < 16; ++q2) s += c0[q2] + c1[q2] + d0[q2] + d1[q2];
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
#define CASE(F) case F: hipLaunchKernelGGL((mix<F>), dim3(256), dim3(256), 49152, (hipStream_t)stream, (const i32x4*)in, out, iters, cyc); break;
extern "C" int mix_launch(const void* in, float* out, int iters, int flags, long long* cyc, void* stream) {
    switch (flags) { CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(15) default: return 2; }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
