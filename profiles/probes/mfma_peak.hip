// scratch: practical peak of v_mfma_f32_32x32x2_f32 (no memory traffic), to calibrate the roofline denominator
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_spin(float* out, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int q = 0; q < 16; ++q) acc[a][q] = 0.f;
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int q = 0; q < 16; ++q) s += acc[a][q];
    if (s == 123.456f) out[0] = s;
}
extern "C" int mfma_spin_launch(float* out, int blocks, int iters, int nacc, void* stream) {
    if (nacc == 4) hipLaunchKernelGGL(mfma_spin<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
    else hipLaunchKernelGGL(mfma_spin<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
