// Write-pattern microbenchmark: how fast can 184 MB of fp32 [2*4800, 4800] be written when every workgroup writes
// (TR rows x TCB bytes) tiles with the volume kernel's store shapes?  Answers "what does the epilogue cost by itself".
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// mode 0: dword stores, a wave instr = 2 rows x 128 B (MFMA 32x32 C layout), tile = 128 rows x (64*NCB) cols, persistent run
// mode 1: same with plain (non-NT) stores
// mode 2: dwordx4 stores, a wave instr = 2 rows x 512 B?  -> lane l: row = l >> 5, 16 B at (l & 31) * 16 : 512 B per row
template <int MODE, int NCB>
__global__ __launch_bounds__(256) void wr(float* __restrict__ out, int N1, int N2, int B, float v) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, kh = lane >> 5, li = lane & 31;
    const int nb = (N1 + 127) >> 7, nc = N2 / (64 * NCB);
    const long per = (long)nb * nc, T = (long)B * per, G = gridDim.x;
    long it = (long)blockIdx.x * T / G;
    const long it_end = ((long)blockIdx.x + 1) * T / G;
    for (; it < it_end; ++it) {
        const int b = (int)(it / per);
        const long r = it - (long)b * per;
        const int band = (int)(r / nc), c = (int)(r - (long)band * nc);
        if (MODE <= 1 || MODE == 4) {
            float* O = out + (size_t)b * N1 * N2 + (size_t)(c * 64 * NCB + li);
            const int row0 = band * 128 + wave * 32 + 4 * kh;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = row0 + (q & 3) + 8 * (q >> 2);
                if (row < N1) {
#pragma unroll
                    for (int j = 0; j < 2 * NCB; ++j) {
                        if (MODE == 0) __builtin_nontemporal_store(v + q, O + (size_t)row * N2 + 32 * j);
                        else if (MODE == 4) {
                            unsigned h = (unsigned)(row * 4801 + c * 64 + li + 32 * j) * 2654435761u;
                            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
                            O[(size_t)row * N2 + 32 * j] = __uint_as_float((h & 0x007fffffu) | 0x3f800000u) - 1.5f;
                        }
                        else O[(size_t)row * N2 + 32 * j] = v + q;
                    }
                }
            }
        } else {
            // 128 rows x 64*NCB cols; wave handles 32 rows; instr covers (64*16 B) / (256*NCB B per row) rows
            constexpr int LPR = 16 * NCB;                 // lanes per row (16 B each)
            constexpr int RPI = 64 / LPR;                 // rows per instruction
            const int lrow = lane / LPR, lcol = lane % LPR;
            float* O = out + (size_t)b * N1 * N2 + (size_t)(c * 64 * NCB + lcol * 4);
#pragma unroll
            for (int q = 0; q < 32 / RPI; ++q) {
                const int row = band * 128 + wave * 32 + q * RPI + lrow;
                if (row < N1) {
                    f32x4 x = {v, v + q, v, v};
                    if (MODE == 2) __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(O + (size_t)row * N2));
                    else *reinterpret_cast<f32x4*>(O + (size_t)row * N2) = x;
                }
            }
        }
    }
}
// mode-1 stores + RD x 4 KB of reads per item and workgroup from a buffer of `rd_lines` 128-B lines (power of two), line picked by a
// hash of (item, j): small buffer = L2 hits, 32 MB = L2 misses / Infinity-Cache hits, 2 GB = HBM reads
template <int RD>
__global__ __launch_bounds__(256) void wr_rd(float* __restrict__ out, const f32x4* __restrict__ src, unsigned rd_mask, int N1, int N2, int B, float v) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, kh = lane >> 5, li = lane & 31;
    const int nb = (N1 + 127) >> 7, nc = N2 / 64;
    const long per = (long)nb * nc, T = (long)B * per, G = gridDim.x;
    long it = (long)blockIdx.x * T / G;
    const long it_end = ((long)blockIdx.x + 1) * T / G;
    f32x4 acc = {0, 0, 0, 0};
    for (; it < it_end; ++it) {
        const int b = (int)(it / per);
        const long r = it - (long)b * per;
        const int band = (int)(r / nc), c = (int)(r - (long)band * nc);
#pragma unroll
        for (int j = 0; j < RD; ++j) {           // 256 threads x 16 B = 4 KB = 32 lines per j; 8 lanes share a line
            unsigned h = ((unsigned)it * 64u + j) * 2654435761u;
            h ^= h >> 13;
            const unsigned line = (h + (t >> 3)) & rd_mask;
            acc += src[(size_t)line * 8 + (t & 7)];
        }
        float* O = out + (size_t)b * N1 * N2 + (size_t)(c * 64 + li);
        const int row0 = band * 128 + wave * 32 + 4 * kh;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = row0 + (q & 3) + 8 * (q >> 2);
            if (row < N1) {
                O[(size_t)row * N2] = v + q;
                O[(size_t)row * N2 + 32] = v + q;
            }
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}
extern "C" int run_rd(int rd, float* out, const void* src, unsigned rd_mask, int N1, int N2, int B, int grid, hipStream_t s) {
#define LR(R) hipLaunchKernelGGL((wr_rd<R>), dim3(grid), dim3(256), 0, s, out, (const f32x4*)src, rd_mask, N1, N2, B, 1.f)
    if (rd == 0) LR(0); else if (rd == 1) LR(1); else if (rd == 2) LR(2); else if (rd == 4) LR(4); else if (rd == 8) LR(8); else if (rd == 16) LR(16); else return -1;
    return (int)hipGetLastError();
}
extern "C" int run(int mode, int ncb, float* out, int N1, int N2, int B, int grid, hipStream_t s) {
#define L(M, C) hipLaunchKernelGGL((wr<M, C>), dim3(grid), dim3(256), 0, s, out, N1, N2, B, 1.f)
    if (mode == 0 && ncb == 1) L(0, 1); else if (mode == 0 && ncb == 2) L(0, 2); else if (mode == 0 && ncb == 4) L(0, 4);
    else if (mode == 1 && ncb == 1) L(1, 1); else if (mode == 1 && ncb == 2) L(1, 2);
    else if (mode == 2 && ncb == 1) L(2, 1); else if (mode == 2 && ncb == 2) L(2, 2); else if (mode == 2 && ncb == 4) L(2, 4);
    else if (mode == 4 && ncb == 1) L(4, 1); else if (mode == 4 && ncb == 2) L(4, 2); else if (mode == 3 && ncb == 1) L(3, 1); else if (mode == 3 && ncb == 4) L(3, 4);
    else return -1;
    return (int)hipGetLastError();
}
