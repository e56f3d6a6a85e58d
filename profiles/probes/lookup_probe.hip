// scratch: tiling variants of the shipped lookup kernel (includes the real source)
#include "../../mac-vo_amd/csrc/corr_lookup.hip"
__global__ void empty_kernel(float* out) { if (out == nullptr) __builtin_trap(); }
extern "C" int probe_launch(const float* vol, const float* coords, float* out, int B, int N1, int H2, int W2, int variant,
                            void* stream) {
    hipStream_t s = (hipStream_t)stream;
#define L(QPW, QPB) hipLaunchKernelGGL((corr_lookup_kernel<4, QPW, QPB>), dim3((N1 + QPB - 1) / QPB, B), dim3(64 * (QPB / QPW)), 0, s, vol, coords, out, N1, H2, W2)
    switch (variant) {
        case 0: hipLaunchKernelGGL(empty_kernel, dim3((N1 + 31) / 32, B), dim3(1024), 0, s, out); break;
        case 1: L(2, 32); break;
        case 2: L(2, 16); break;
        case 3: L(1, 16); break;
        case 4: L(4, 32); break;
        case 5: L(1, 8); break;
        case 6: L(2, 8); break;
        case 7: L(4, 16); break;
        case 8: L(8, 32); break;
        case 9: L(4, 64); break;
        case 10: L(1, 4); break;
        case 11: L(2, 4); break;
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
