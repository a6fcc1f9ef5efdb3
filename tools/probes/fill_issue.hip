// Probe (round 6): what does one k-tile of a 256 x 256 x 64 GEMM cost on FOUR waves per CU (one per SIMD, 128 x 128 wave
// tiles: 128 MFMAs per wave and k-tile) as a function of HOW the 64 KiB of operands reach LDS?  The k-tile is 16 blocks of
// 8 mfma_f32_16x16x32_bf16 (64 accumulators = 256 registers, as hv_gemm_w4_kernel); each block additionally carries
//   MODE 0: nothing (the matrix pipe alone: 16 x 8 x 16 = 2048 cycles)
//   MODE 1: one LDS-DMA piece (global_load_lds_dwordx4, 1 KiB, M0 written in the statement)
//   MODE 2: one LDS-DMA piece as buffer_load_dwordx4 ... offen lds
//   MODE 3: one global_load_dwordx4 into a register ring + the ds_write_b128 of the piece loaded one k-tile earlier
//   MODE 4: two ds_read_b128 whose results feed the block's MFMAs two blocks later (fragment reads alone)
//   MODE 5: MODE 1 + MODE 4 (= the loop of hv_gemm_w4_kernel),  MODE 6: MODE 3 + MODE 4 (register-staged loop)
//   MODE 7: MODE 2 + MODE 4
// One raw barrier per k-tile in every mode.  SRC: 0 = every k-tile re-reads the workgroup's 64 KiB window (L2-hot),
// 1 = a quarter of the pieces stream through a large buffer (fresh lines), the rest hot.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/fill_issue.hip -o tools/bin/fill_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds(const void* base, unsigned ofs, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(ofs), "s"(base), "s"(lds) : "memory");
}
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void blds(i32x4 rsrc, unsigned ofs, unsigned soff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(ofs), "s"(rsrc), "s"(soff), "s"(lds) : "memory");
}

template <int MODE, int SRC>
__global__ __launch_bounds__(256, 1) void probe(const unsigned char* __restrict__ src, float* out, int steps, long stream_stride) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[160 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr bool DMA = MODE == 1 || MODE == 5, BDMA = MODE == 2 || MODE == 7, REG = MODE == 3 || MODE == 6;
    constexpr bool RD = MODE >= 4;
    f32x4 acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 wf[8], xr[4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[i][j] = (short)(0x3c00 + ((tid + i + j) & 63));
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) xr[i][j] = (short)(0x3c00 + ((tid * 3 + i + j) & 63));
    for (int i = tid; i < 160 * 1024 / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + (i & 15);
    __syncthreads();
    const unsigned char* hot = src + (long)blockIdx.x * 65536;
    const unsigned char* cold = src + 256L * 65536 + (long)blockIdx.x * stream_stride;
    const unsigned lofs = (unsigned)lane * 16u;
    i32x4 rsrc;
    {
        const unsigned long long a = (unsigned long long)src;
        rsrc[0] = (int)(a & 0xffffffffu);
        rsrc[1] = (int)((a >> 32) & 0xffffu);
        rsrc[2] = -1;
        rsrc[3] = 0x00020000;
    }
    u32x4 stage[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) stage[i] = u32x4{0u, 0u, 0u, 0u};
    unsigned rdo = (unsigned)((lane & 15) * 128 + (lane >> 4) * 16 + wave * 8192);
    int slot = 0;
    for (int s = 0; s < steps; ++s) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned sl = (unsigned)slot * 65536u;
        slot ^= 1;
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const bool fresh = SRC == 1 && (b & 3) == 0;
            const unsigned char* base = fresh ? cold + (long)s * 16384 + (b >> 2) * 4096 + wave * 1024 : hot + (b * 4 + wave) * 1024;
            if (RD) {
                xr[(b + 2) & 3] = *reinterpret_cast<const bf16x8*>(smem + ((sl ^ 65536u) + rdo + ((b * 2048) & 32767)));
                if (b < 8) wf[b] = *reinterpret_cast<const bf16x8*>(smem + ((sl ^ 65536u) + 32768u + rdo + ((b * 2048) & 32767)));
            }
            if (DMA) glds(base, lofs, sl + (unsigned)(b * 4 + wave) * 1024u);
            if (BDMA) blds(rsrc, lofs, (unsigned)(base - src), sl + (unsigned)(b * 4 + wave) * 1024u);
            if (REG) {
                *reinterpret_cast<u32x4*>(smem + sl + (unsigned)(b * 4 + wave) * 1024u + lofs) = stage[b];
                stage[b] = *reinterpret_cast<const u32x4*>(base + lofs);
            }
#pragma unroll
            for (int nf = 0; nf < 8; ++nf)
                acc[nf * 8 + (b & 7)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nf], xr[b & 3], acc[nf * 8 + (b & 7)], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 16; ++i) r += (float)stage[i][0];
    out[blockIdx.x * 256 + tid] = r;
}

template <int MODE, int SRC>
void run(const char* name, const unsigned char* src, float* out, long stride, int steps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    probe<MODE, SRC><<<256, 256>>>(src, out, 20, stride);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        probe<MODE, SRC><<<256, 256>>>(src, out, steps, stride);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double ns = best * 1e6 / steps;
    printf("%-86s %8.1f ns per k-tile  (%6.0f TF/s over 256 CUs)\n", name, ns, 2.0 * 256 * 256 * 64 * 256 / ns / 1e3);
    fflush(stdout);
}

int main() {
    const int steps = 2000;
    const long stride = (long)steps * 16384 + 65536;
    const size_t bytes = 256L * 65536 + 256L * stride + (1 << 20);
    unsigned char* src;
    float* out;
    if (hipMalloc(&src, bytes) != hipSuccess) return 1;
    hipMemset(src, 0x3c, bytes);
    hipMalloc(&out, 256 * 256 * 4);
    hipDeviceSynchronize();
    run<0, 0>("0: MFMAs only", src, out, stride, steps);
    run<4, 0>("4: + 2 ds_read_b128 per block (fragment reads)", src, out, stride, steps);
    run<1, 0>("1: + 1 LDS-DMA piece per block (global_load_lds_dwordx4), L2-hot", src, out, stride, steps);
    run<2, 0>("2: + 1 LDS-DMA piece per block (buffer_load_dwordx4 offen lds), L2-hot", src, out, stride, steps);
    run<3, 0>("3: + 1 global_load_dwordx4 + 1 ds_write_b128 per block (register-staged), L2-hot", src, out, stride, steps);
    run<5, 0>("5: LDS-DMA (global) + fragment reads = the hv_gemm_w4_kernel loop, L2-hot", src, out, stride, steps);
    run<7, 0>("7: LDS-DMA (buffer) + fragment reads, L2-hot", src, out, stride, steps);
    run<6, 0>("6: register-staged + fragment reads, L2-hot", src, out, stride, steps);
    run<5, 1>("5: LDS-DMA (global) + fragment reads, a quarter of the pieces fresh from HBM", src, out, stride, steps);
    run<6, 1>("6: register-staged + fragment reads, a quarter of the pieces fresh from HBM", src, out, stride, steps);
    run<0, 0>("0: MFMAs only (again)", src, out, stride, steps);
    return 0;
}
