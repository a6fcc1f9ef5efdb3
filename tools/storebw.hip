// Store-pattern probe: each wave writes a 128 x 64 bf16 sub-tile of a row-major [M][N] matrix (N = 960) per "tile",
// like the GEMM epilogue, with (A) the MFMA-native pattern: 8 B per lane, 16 rows x 32 B per instruction,
// (B) 16 B per lane, 8 rows x 128 B per instruction, (C) 16 B per lane, 4 rows x 256 B... (whole 128-col tile rows).
// Build: hipcc --offload-arch=gfx950 -O3 tools/storebw.hip -o tools/bin/storebw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void st(char* __restrict__ Y, int M, int N, int tiles_per_block) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r16 = lane & 15, quad = lane >> 4;
    const int tiles_n = N / 128;
    for (int t = 0; t < tiles_per_block; ++t) {
        const int tile = blockIdx.x + t * gridDim.x;
        const int m0 = (tile / tiles_n) * 256 + (wave & 1) * 128, n0 = (tile % tiles_n) * 128 + (wave >> 1) * 64;
        if (m0 >= M) break;
        if (MODE == 0) {
#pragma unroll
            for (int mf = 0; mf < 8; ++mf)
#pragma unroll
                for (int nf = 0; nf < 4; ++nf) {
                    u32x2 v = {(unsigned)t, (unsigned)lane};
                    *reinterpret_cast<u32x2*>(Y + ((long)(m0 + 16 * mf + r16) * N + n0 + 16 * nf + 4 * quad) * 2) = v;
                }
        } else if (MODE == 3) {
            // (D) the round-2 candidate: 16 B per lane, 16 rows x 64 B per instruction (two adjacent MFMA fragments per lane)
#pragma unroll
            for (int mf = 0; mf < 8; ++mf)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    u32x4 v = {(unsigned)t, (unsigned)lane, 0u, 1u};
                    *reinterpret_cast<u32x4*>(Y + ((long)(m0 + 16 * mf + r16) * N + n0 + 32 * j + 8 * quad) * 2) = v;
                }
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                u32x4 v = {(unsigned)t, (unsigned)lane, 0u, 1u};
                *reinterpret_cast<u32x4*>(Y + ((long)(m0 + 8 * i + (lane >> 3)) * N + n0 + (lane & 7) * 8) * 2) = v;
            }
        } else {
            // wave owns 32 full tile rows (256 B each): 16 lanes x 16 B per row, 4 rows per instruction
            const int mw = (tile / tiles_n) * 256 + wave * 64, nw = (tile % tiles_n) * 128;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                u32x4 v = {(unsigned)t, (unsigned)lane, 0u, 1u};
                *reinterpret_cast<u32x4*>(Y + ((long)(mw + 4 * i + (lane >> 4)) * N + nw + (lane & 15) * 8) * 2) = v;
            }
        }
    }
}

int main() {
    const int M = 294912;
    char* Y;
    hipMalloc(&Y, (size_t)M * 1280 * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* names[] = {"A: 8 B/lane, 16 rows x 32 B per instr (MFMA native)", "B: 16 B/lane, 8 rows x 128 B per instr",
                           "C: 16 B/lane, 4 rows x 256 B per instr", "D: 16 B/lane, 16 rows x 64 B per instr"};
    const int Ns[] = {960, 1280, 640, 2560, 5120, 320};
    for (int N : Ns) {
        const int Mx = N <= 1280 ? M : (N == 2560 ? M / 4 : M / 16);  // level-1 / level-2 row counts
        const int tiles = (Mx / 256) * (N / 128), grid = 512, tpb = (tiles + grid - 1) / grid;
        for (int mode = 0; mode < 4; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(st<0>, dim3(grid), dim3(256), 0, 0, Y, Mx, N, tpb);
                if (mode == 1) hipLaunchKernelGGL(st<1>, dim3(grid), dim3(256), 0, 0, Y, Mx, N, tpb);
                if (mode == 2) hipLaunchKernelGGL(st<2>, dim3(grid), dim3(256), 0, 0, Y, Mx, N, tpb);
                if (mode == 3) hipLaunchKernelGGL(st<3>, dim3(grid), dim3(256), 0, 0, Y, Mx, N, tpb);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            printf("N=%5d M=%6d %-52s %7.3f ms  %6.2f TB/s\n", N, Mx, names[mode], ms, (double)Mx * N * 2 / ms / 1e9);
        }
    }
    return 0;
}
