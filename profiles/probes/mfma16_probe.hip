// scratch: issue rate of v_mfma_f32_32x32x16_bf16 with ONE wave per SIMD, by number of independent accumulator chains and by
// where the A operand lives (VGPR / AGPR), random or zero data.  No memory traffic in the loop.
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC, bool A_IN_AGPR, int NA>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void spin(const i32x4* in, float* out, int iters, long long* cyc) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int q = 0; q < 16; ++q) acc[a][q] = 0.f;
    i32x4 av[NA], bv[2];
    for (int i = 0; i < NA; ++i) av[i] = in[(threadIdx.x + 64 * i) & 1023];
    bv[0] = in[(threadIdx.x + 7) & 1023];
    bv[1] = in[(threadIdx.x + 77) & 1023];
    if (A_IN_AGPR) { for (int i = 0; i < NA; ++i) asm volatile("" : "+a"(av[i])); }
    else { for (int i = 0; i < NA; ++i) asm volatile("" : "+v"(av[i])); }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < NA; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a)
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[u]), __builtin_bit_cast(bf16x8, bv[a & 1]), acc[a], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int q = 0; q < 16; ++q) s += acc[a][q];
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
#define L(N, AG, NA) hipLaunchKernelGGL((spin<N, AG, NA>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const i32x4*)in, out, iters, cyc)
extern "C" int spin_launch(const void* in, float* out, int blocks, int iters, int nacc, int agpr, long long* cyc, void* stream) {
    if (agpr == 0) { if (nacc == 1) L(1, false, 8); else if (nacc == 2) L(2, false, 8); else if (nacc == 4) L(4, false, 8); else L(8, false, 8); }
    else if (agpr == 1) { if (nacc == 1) L(1, true, 8); else if (nacc == 2) L(2, true, 8); else if (nacc == 4) L(4, true, 8); else L(8, true, 8); }
    else { if (nacc == 2) L(2, true, 48); else L(4, true, 48); }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
