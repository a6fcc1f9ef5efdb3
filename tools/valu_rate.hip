// Issue cost of the softmax instructions of the attention kernels on one SIMD, alone and beside MFMAs (gfx950):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/valu_rate.hip -o tools/bin/valu_rate && tools/bin/valu_rate
// One workgroup of W waves on one CU (W = 4: one wave per SIMD, 8: two, 16: four); every wave runs REP iterations of an
// unrolled block of N independent instructions of one kind between two s_memtime reads; prints shader cycles per
// wave-instruction per SIMD (= cycles x SIMDs / instructions issued on the CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ void rate_kernel(unsigned long long* out, int rep, float seed) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed + threadIdx.x * 1e-3f + i;
    unsigned u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) u[i] = threadIdx.x * 16 + i;
    f32x16 acc, acc2;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = acc2[i] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = b[i] = (short)(0x3c00 + threadIdx.x);
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 o4[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) o4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) {
        if (KIND == 0) {  // 16 x v_exp_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        } else if (KIND == 1) {  // 16 x v_mul_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(v[i]));
        } else if (KIND == 2) {  // 8 x v_cvt_pk_bf16_f32
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(v[2 * i]), "v"(v[2 * i + 1]));
        } else if (KIND == 3) {  // 8 x v_permlane16_swap
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(u[2 * i]), "+v"(u[2 * i + 1]));
        } else if (KIND == 4) {  // 8 x v_pk_mul_f32
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(*reinterpret_cast<double*>(&v[2 * i])));
        } else if (KIND == 5) {  // 2 x mfma 32x32x16 alone
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
        } else if (KIND == 6) {  // 2 x mfma 32x32x16, ONE accumulation chain
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        } else if (KIND == 7 || KIND == 8 || KIND == 9) {  // 6 x mfma 16x16x32 over 1 / 2 / 6 accumulation chains
            constexpr int NC = KIND == 7 ? 1 : (KIND == 8 ? 2 : 6);
#pragma unroll
            for (int i = 0; i < 6; ++i) o4[i % NC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, o4[i % NC], 0, 0, 0);
        } else if (KIND >= 10 && KIND < 20) {  // 2 x mfma 32x32x16 with (KIND - 10) v_exp_f32 behind each
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < KIND - 10; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < KIND - 10; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[8 + i]));
        } else if (KIND >= 20 && KIND < 30) {  // 2 x mfma with (KIND - 20) v_mul_f32 behind each
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < KIND - 20; ++i) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(v[i]));
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < KIND - 20; ++i) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(v[8 + i]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i] + acc[i] + acc2[i] + (float)u[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) s += o4[i][0] + o4[i][1] + o4[i][2] + o4[i][3];
    if (threadIdx.x % 64 == 0) out[threadIdx.x / 64] = t1 - t0;
    if (s == 12345.678f) out[63] = 1;
}

// waves of a SIMD in different phases: even waves (of the SIMD) run MFMAs while odd waves run exps -- the cross-wave overlap
// the 4-waves-per-SIMD attention kernel relies on.  A wave's SIMD is (wave id % 4) on gfx950's cyclic dispatch, so waves
// w and w + 4 share one: w < 4 -> MFMA role, w >= 4 -> exp role.
__global__ void mix_kernel(unsigned long long* out, int rep, int n_mfma, int n_exp) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
    f32x16 acc, acc2;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = acc2[i] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = b[i] = (short)(0x3c00 + threadIdx.x);
    const int wave = threadIdx.x / 64;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        for (int r = 0; r < rep * n_mfma; ++r) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
        }
    } else {
        for (int r = 0; r < rep * n_exp; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i] + acc[i] + acc2[i];
    if (threadIdx.x % 64 == 0) out[wave] = t1 - t0;
    if (s == 12345.678f) out[63] = 1;
}


// The instruction stream of one hv_attention40 wave-tile without its memory side: 6 MFMA 32x32x16 (two chains of 3), 32 v_exp_f32
// on their results, 16 packs, 8 lane-row swaps, 12 MFMA 16x16x32 on the swapped pairs.  One workgroup of 8 or 16 waves per CU
// (two or four such waves per SIMD), optionally a workgroup barrier per iteration.  Pipe work per wave-iteration:
// 384 matrix cycles, ~410 VALU cycles.
template <int BAR, int PART>
__global__ __launch_bounds__(1024, 4) void attn_like_kernel(unsigned long long* out, int rep) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = b[i] = (short)(0x3c00 + threadIdx.x);
    f32x4 o4[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) o4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) {
        f32x16 s0 = z, s1 = z;
        if (PART & 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 3; ++k) s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s1, 0, 0, 0);
        }
        unsigned w[2][8];
        if (PART & 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(s0[i]));
                asm volatile("v_exp_f32 %0, %0" : "+v"(s1[i]));
            }
        }
        if (PART & 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[0][j]) : "v"(s0[2 * j]), "v"(s0[2 * j + 1]));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[1][j]) : "v"(s1[2 * j]), "v"(s1[2 * j + 1]));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) w[0][j] = __builtin_bit_cast(unsigned, s0[j]), w[1][j] = __builtin_bit_cast(unsigned, s1[j]);
        }
        if (PART & 16) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int d = 0; d < 4; ++d) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(w[k][d]), "+v"(w[k][d + 4]));
        }
        if (PART & 4) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const bf16x8 p0 = __builtin_bit_cast(bf16x8, u32x4{w[k][0], w[k][1], w[k][2], w[k][3]});
                const bf16x8 p1 = __builtin_bit_cast(bf16x8, u32x4{w[k][4], w[k][5], w[k][6], w[k][7]});
#pragma unroll
                for (int dt = 0; dt < 3; ++dt) {
                    o4[2 * dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, p0, o4[2 * dt], 0, 0, 0);
                    o4[2 * dt + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, p1, o4[2 * dt + 1], 0, 0, 0);
                }
            }
        } else {
            o4[0][0] += __builtin_bit_cast(float, w[0][0] ^ w[1][7] ^ w[0][4] ^ w[1][3]);
        }
        if (BAR) __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) sum += o4[i][0] + o4[i][1] + o4[i][2] + o4[i][3];
    if (threadIdx.x % 64 == 0 && blockIdx.x == 0) out[threadIdx.x / 64] = t1 - t0;
    if (sum == 12345.678f) out[63] = 1;
}

template <int BAR, int PART>
static void run_attn_like(const char* what, int blocks_per_cu) {
    unsigned long long* d;
    hipMalloc(&d, 64 * 8);
    hipMemset(d, 0, 64 * 8);
    const int rep = 2048;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int grid = prop.multiProcessorCount;
    attn_like_kernel<BAR, PART><<<grid, 512 * blocks_per_cu>>>(d, rep);
    attn_like_kernel<BAR, PART><<<grid, 512 * blocks_per_cu>>>(d, rep);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(64);
    hipMemcpy(h.data(), d, 64 * 8, hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (int w = 0; w < 8 * blocks_per_cu; ++w) mx = h[w] > mx ? h[w] : mx;
    printf("%-58s %d x 8 waves per CU: %7.1f cycles per iteration = %6.1f per wave-iteration and SIMD\n", what, blocks_per_cu,
           (double)mx / rep, (double)mx / rep / (2.0 * blocks_per_cu));
    hipFree(d);
}


// Do transcendentals and plain VALU of two waves of one SIMD overlap?  Waves 0-3 run v_exp_f32, waves 4-7 (same SIMDs) ROLE_B:
// 0 nothing, 1 v_mul_f32, 2 v_pk_fma_f32, 3 v_exp_f32.
template <int ROLE_B>
__global__ void trans_mix_kernel(unsigned long long* out, int rep) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.001f * threadIdx.x + i * 0.01f;
    const int wave = threadIdx.x / 64;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4 || ROLE_B == 3) {
        for (int r = 0; r < rep; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %1" : "=v"(v[i]) : "v"(v[(i + 1) & 15]));
        }
    } else if (ROLE_B == 1) {
        for (int r = 0; r < rep * 3; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(v[i]));
        }
    } else if (ROLE_B == 2) {
        for (int r = 0; r < rep * 2; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*reinterpret_cast<double*>(&v[2 * i])));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += v[i];
    if (threadIdx.x % 64 == 0) out[wave] = t1 - t0;
    if (sum == 12345.678f) out[63] = 1;
}
template <int ROLE_B>
static void run_trans_mix(const char* what, int per_b) {
    unsigned long long* d;
    hipMalloc(&d, 64 * 8);
    const int rep = 2048;
    trans_mix_kernel<ROLE_B><<<1, 512>>>(d, rep);
    trans_mix_kernel<ROLE_B><<<1, 512>>>(d, rep);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(64);
    hipMemcpy(h.data(), d, 64 * 8, hipMemcpyDeviceToHost);
    printf("wave A v_exp_f32 beside wave B %-14s: A %.2f cycles per v_exp_f32", what, (double)h[0] / rep / 16);
    if (per_b) printf(", B %.2f cycles per instruction", (double)h[4] / rep / per_b);
    printf("\n");
    hipFree(d);
}

template <int KIND>
static void run(const char* what, int per_iter, int waves) {
    unsigned long long* d;
    hipMalloc(&d, 64 * 8);
    const int rep = 4096;
    rate_kernel<KIND><<<1, waves * 64>>>(d, rep, 0.5f);
    rate_kernel<KIND><<<1, waves * 64>>>(d, rep, 0.5f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(64);
    hipMemcpy(h.data(), d, 64 * 8, hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
    const double per_simd = (double)waves / 4 * rep * per_iter;
    printf("%-44s %2d waves/CU: %6.2f cycles per wave-instruction per SIMD (block of %d: %.1f)\n", what, waves, mx / per_simd, per_iter,
           mx / per_simd * per_iter);
    hipFree(d);
}

int main() {
    for (int waves : {4, 8, 16}) {
        run<0>("v_exp_f32", 16, waves);
        run<1>("v_mul_f32", 16, waves);
        run<2>("v_cvt_pk_bf16_f32", 8, waves);
        run<3>("v_permlane16_swap_b32", 8, waves);
        run<4>("v_pk_mul_f32", 8, waves);
        run<5>("v_mfma_f32_32x32x16_bf16, two chains", 2, waves);
        run<6>("v_mfma_f32_32x32x16_bf16, one chain", 2, waves);
        run<7>("v_mfma_f32_16x16x32_bf16, one chain", 6, waves);
        run<8>("v_mfma_f32_16x16x32_bf16, two chains", 6, waves);
        run<9>("v_mfma_f32_16x16x32_bf16, six chains", 6, waves);
    }
    for (int waves : {4, 8}) {
        run<12>("mfma + 2 v_exp_f32 (cycles per MFMA)", 2, waves);
        run<14>("mfma + 4 v_exp_f32 (cycles per MFMA)", 2, waves);
        run<16>("mfma + 6 v_exp_f32 (cycles per MFMA)", 2, waves);
        run<18>("mfma + 8 v_exp_f32 (cycles per MFMA)", 2, waves);
        run<24>("mfma + 4 v_mul_f32 (cycles per MFMA)", 2, waves);
        run<26>("mfma + 6 v_mul_f32 (cycles per MFMA)", 2, waves);
        run<28>("mfma + 8 v_mul_f32 (cycles per MFMA)", 2, waves);
    }
    for (int bpc : {1, 2}) {
        if (bpc == 1) {
            run_attn_like<0, 26>("attention-like wave: softmax VALU only", 1);
            run_attn_like<0, 4>("attention-like wave: PV MFMAs only", 1);
            run_attn_like<0, 31>("attention-like wave: all", 1);
            run_attn_like<1, 31>("attention-like wave: all + barrier per iteration", 1);
        } else {
            run_attn_like<0, 26>("attention-like wave: softmax VALU only", 2);
            run_attn_like<0, 2>("attention-like wave: 32 v_exp_f32 only", 2);
            run_attn_like<0, 8>("attention-like wave: 16 v_cvt_pk_bf16_f32 only", 2);
            run_attn_like<0, 16>("attention-like wave: 8 v_permlane16_swap only", 2);
            run_attn_like<0, 5>("attention-like wave: MFMAs only (18)", 2);
            run_attn_like<0, 7>("attention-like wave: MFMAs + exps", 2);
            run_attn_like<0, 15>("attention-like wave: MFMAs + exps + packs", 2);
            run_attn_like<0, 31>("attention-like wave: all", 2);
            run_attn_like<1, 31>("attention-like wave: all + barrier per iteration", 2);
        }
    }
    run_trans_mix<0>("(idle)", 0);
    run_trans_mix<1>("v_mul_f32", 48);
    run_trans_mix<2>("v_pk_fma_f32", 16);
    run_trans_mix<3>("v_exp_f32", 16);
    // cross-wave: waves 0-3 run 2 n MFMAs per round, waves 4-7 (same SIMDs) 16 m exps per round
    for (int m : {0, 2, 4, 8}) {
        unsigned long long* d;
        hipMalloc(&d, 64 * 8);
        const int rep = 1024, n = 4;
        mix_kernel<<<1, 512>>>(d, rep, n, m);
        mix_kernel<<<1, 512>>>(d, rep, n, m);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(64);
        hipMemcpy(h.data(), d, 64 * 8, hipMemcpyDeviceToHost);
        printf("cross-wave: wave A %d MFMAs, wave B %d v_exp_f32 per round: A %.0f cycles per round (alone %d), B %.0f\n", 2 * n, 16 * m,
               (double)h[0] / rep, 64 * n, (double)h[4] / rep);
        hipFree(d);
    }
    return 0;
}
