// Probe: do MFMA and VALU work of DIFFERENT waves on one SIMD overlap, and does the phase order inside a wave matter?
// Each wave runs ITERS iterations of "a tile": NM mfma_f32_16x16x32_bf16 (8 independent accumulator chains) and a VALU block
// of NE v_exp_f32 + NF v_fma_f32 on independent registers.  MODE 0 = MFMA only, 1 = VALU only, 2 = MFMA block then VALU
// block (the attention kernel's shape), 3 = interleaved (one MFMA, then NE/NM exps + NF/NM fmas).  Occupancy 1 / 2 / 4 waves
// per SIMD via the grid (256-thread blocks = one wave per SIMD; blocks per CU = waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap.hip -o tools/bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE, int NM, int NE, int NF>
__global__ __launch_bounds__(256, 4) void probe(float* out, int iters, unsigned long long* cyc) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)(0.001f * (threadIdx.x + i)), b[i] = (__bf16)(0.002f * (threadIdx.x - i));
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float e[32], f[32];
    for (int i = 0; i < 32; ++i) e[i] = -0.001f * (threadIdx.x + i), f[i] = 0.5f + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[m % 8] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m % 8], 0, 0, 0);
        }
        if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < NE; ++i) e[i % 32] = __builtin_amdgcn_exp2f(e[i % 32]);
#pragma unroll
            for (int i = 0; i < NF; ++i) f[i % 32] = __builtin_fmaf(f[i % 32], 0.999f, 0.001f);
        }
        if (MODE == 3) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                acc[m % 8] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m % 8], 0, 0, 0);
#pragma unroll
                for (int i = m * NE / NM; i < (m + 1) * NE / NM; ++i) e[i % 32] = __builtin_amdgcn_exp2f(e[i % 32]);
#pragma unroll
                for (int i = m * NF / NM; i < (m + 1) * NF / NM; ++i) f[i % 32] = __builtin_fmaf(f[i % 32], 0.999f, 0.001f);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 32; ++i) s += e[i] + f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NM, int NE, int NF>
void run(const char* name, int wps) {
    const int iters = 2000, blocks = 256 * wps;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, blocks * 256 * 4);
    hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    probe<MODE, NM, NE, NF><<<blocks, 256>>>(out, 10, cyc);
    hipEventRecord(e0);
    probe<MODE, NM, NE, NF><<<blocks, 256>>>(out, iters, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024];
    hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < blocks; ++i) mean += h[i];
    mean /= blocks;
    // s_memtime ticks at 100 MHz on gfx9: convert wall time with the event time instead
    printf("%-44s waves/SIMD %d: %8.3f ms  -> %7.1f ns per tile per SIMD | s_memtime %8.0f ticks per tile and wave, %7.1f per SIMD tile (%.2f ticks/ns)\n",
           name, wps, ms, ms * 1e6 / iters / wps, mean / iters, mean / iters / wps, mean / (ms * 1e6));
    hipFree(out), hipFree(cyc);
}

int main() {
    for (int wps : {1, 4}) {
        run<0, 28, 32, 60>("MFMA only (28 x 16x16x32)", wps);
        run<1, 28, 32, 60>("VALU only (32 exp + 60 fma)", wps);
        run<1, 28, 32, 0>("VALU only (32 exp)", wps);
        run<1, 28, 0, 60>("VALU only (60 fma)", wps);
        run<2, 28, 32, 60>("MFMA block, then VALU block", wps);
        run<3, 28, 32, 60>("interleaved MFMA / VALU", wps);
        run<2, 28, 0, 60>("MFMA block, then 60 fma", wps);
        run<3, 28, 0, 60>("interleaved MFMA / 60 fma", wps);
        run<3, 28, 0, 28>("interleaved MFMA / 28 fma", wps);
        run<3, 28, 16, 0>("interleaved MFMA / 16 exp", wps);
        run<2, 28, 32, 0>("MFMA block, then 32 exp", wps);
        run<3, 28, 32, 0>("interleaved MFMA / 32 exp", wps);
    }
    return 0;
}
