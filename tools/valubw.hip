// VALU issue-rate probe: is v_exp_f32 (transcendental) serialised with v_fma_f32 / v_pk_fma_f32 on a SIMD, and what
// are their rates?  Each wave runs a long unrolled loop of independent ops; 4 waves per SIMD (1024 blocks x 256 thr).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valubw.hip -o tools/bin/valubw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NEXP, int NFMA, int NPK>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float e[8], f[8];
    f32x2 pk[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = seed * (i + 1) * 1e-3f - threadIdx.x * 1e-4f, f[i] = seed + i, pk[i] = f32x2{seed + i, seed - i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NEXP; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(e[i % 8]));
#pragma unroll
        for (int i = 0; i < NFMA; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i % 8]) : "v"(seed));
#pragma unroll
        for (int i = 0; i < NPK; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pk[i % 8]) : "v"(pk[(i + 1) % 8]));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += e[i] + f[i] + pk[i][0] + pk[i][1];
    if (s == 1.2345f) out[0] = s;
}

template <int NEXP, int NFMA, int NPK>
void run(const char* name, float* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    float ms = 0;
    for (int r = 0; r < 2; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NEXP, NFMA, NPK>), dim3(1024), dim3(256), 0, 0, d, iters, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    // per SIMD: 4 waves x iters x (ops); report cycles per wave-instruction group assuming 2.4 GHz
    const double waves_per_simd = 1024.0 * 4 / (256 * 4);
    const double cyc = ms * 1e-3 * 2.4e9 / (iters * waves_per_simd);
    printf("%-44s %8.3f ms  %7.1f cycles per wave-iteration (%d exp, %d fma, %d pk_fma)\n", name, ms, cyc, NEXP, NFMA, NPK);
}

int main() {
    float* d;
    hipMalloc(&d, 4);
    run<8, 0, 0>("8 x v_exp_f32", d);
    run<0, 32, 0>("32 x v_fma_f32", d);
    run<0, 0, 32>("32 x v_pk_fma_f32", d);
    run<8, 32, 0>("8 x v_exp_f32 + 32 x v_fma_f32", d);
    run<8, 0, 32>("8 x v_exp_f32 + 32 x v_pk_fma_f32", d);
    return 0;
}
