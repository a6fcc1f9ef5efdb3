// Fill-bandwidth probe: how fast can a CU pull bf16 tiles into LDS (global_load_lds) or registers when the
// source is (a) one hot 24 KB tile (L1/L2 hot), (b) a per-XCD working set that fits L2, (c) a streaming HBM read.
// Prints GB/s per CU and chip TB/s.  Build: hipcc --offload-arch=gfx950 -O3 tools/fillbw.hip -o tools/bin/fillbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int PAT>  // MODE 0 = glds, 1 = register loads; PAT 0 contiguous, 1: 16 rows x 64 B, 2: 8 rows x 128 B (row stride 640 B)
__global__ __launch_bounds__(256, 2) void fill(const char* __restrict__ src, long span_per_block, long block_stride, int share, int iters,
                                               unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const char* base = src + (long)(blockIdx.x / share) * block_stride;
    u32x4 acc = {0, 0, 0, 0};
    long off = 0;
    for (int it = 0; it < iters; ++it) {
        // 24 KB per iteration: 6 x 4 KB (256 threads x 16 B)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int ti = j * 256 + tid;  // 16-byte chunk index inside the 24 KB step
            const char* g = PAT == 0 ? base + off + ti * 16
                          : PAT == 1 ? base + off + (long)(ti / 4) * 640 + (ti % 4) * 16
                                     : base + off + (long)(ti / 8) * 640 + (ti % 8) * 16;
            if (MODE == 0) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(lds + ((it & 1) * 6 + j) * 4096 + wave * 1024),
                                                 16, 0, 0);
            } else {
                acc ^= *reinterpret_cast<const u32x4*>(g);
            }
        }
        off += PAT == 0 ? 24576 : PAT == 1 ? 64 : 128;
        if (PAT == 0 ? off + 24576 > span_per_block : off >= 640) off = 0;
        if (MODE == 0 && (it & 3) == 3) __builtin_amdgcn_s_waitcnt(0x0f70 | 6 | (0 << 14));  // vmcnt(6): keep 6 in flight
    }
    if (MODE == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        acc[0] = reinterpret_cast<unsigned*>(lds)[tid];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345) sink[0] = 1;
}

int main(int argc, char** argv) {
    const long total = 2L << 30;
    char* d;
    unsigned* sink;
    hipMalloc(&d, total);
    hipMemset(d, 1, total);
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    if (argc > 1) {
        // calibration of rocprofv3 FETCH_SIZE (run under --pmc FETCH_SIZE): every byte of a 983 MB region is read
        // exactly once, 4096 workgroups x 240 KB, as contiguous 1 KiB wave reads / as 64-byte row segments (the
        // BK = 32 GEMM tile pattern) / as 128-byte row segments
        hipLaunchKernelGGL((fill<0, 0>), dim3(4096), dim3(256), 49152, 0, d, 245760L, 245760L, 1, 10, sink);
        hipLaunchKernelGGL((fill<0, 1>), dim3(4096), dim3(256), 49152, 0, d, 640L, 245760L, 1, 10, sink);
        hipLaunchKernelGGL((fill<0, 2>), dim3(4096), dim3(256), 49152, 0, d, 640L, 245760L, 1, 5, sink);
        hipDeviceSynchronize();
        printf("calibration: 3 launches, %.1f MB read once each\n", 4096 * 245760.0 / 1e6);
        return 0;
    }
    const int grid = 512;
    struct Case { const char* name; long span, stride; int share, iters; };
    Case cases[] = {
        {"hot: every block re-reads the same 24 KB", 24576, 0, 1, 2000},
        {"L2 working set: 48 KB per block (24 MB total), private", 49152, 49152, 1, 2000},
        {"L2 shared: 8 consecutive blocks share each 192 KB span", 196608, 196608, 8, 2000},
        {"MALL: 384 KB per block private (192 MB total)", 393216, 393216, 1, 1000},
        {"HBM stream: 4 MB per block private, one pass", 4L << 20, 4L << 20, 1, 170},
    };
    for (int mode = 0; mode < 2; ++mode)
        for (auto& c : cases) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0)
                    hipLaunchKernelGGL((fill<0, 0>), dim3(grid), dim3(256), 49152, 0, d, c.span, c.stride, c.share, c.iters, sink);
                else
                    hipLaunchKernelGGL((fill<1, 0>), dim3(grid), dim3(256), 49152, 0, d, c.span, c.stride, c.share, c.iters, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double bytes = (double)grid * c.iters * 24576.0;
            printf("%s %-58s %8.3f ms  %7.1f GB/s/CU  %6.2f TB/s\n", mode == 0 ? "glds" : "regs", c.name, ms,
                   bytes / ms / 1e6 / 256, bytes / ms / 1e9);
        }
    // GEMM-like tiles: each block walks the K range of a private 384-row x 640-B tile (240 KB, L2-resident working set 120 MB -> MALL/L2 mix),
    // and the same with every block on the same tile (L2 hot)
    for (int pat = 1; pat <= 2; ++pat)
        for (int hot = 0; hot < 2; ++hot) {
            const long stride = hot ? 0 : 384 * 640;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (pat == 1)
                    hipLaunchKernelGGL((fill<0, 1>), dim3(grid), dim3(256), 49152, 0, d, 640L, stride, 1, 2000, sink);
                else
                    hipLaunchKernelGGL((fill<0, 2>), dim3(grid), dim3(256), 49152, 0, d, 640L, stride, 1, 2000, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double bytes = (double)grid * 2000 * 24576.0;
            printf("glds rows of %3d B (%2d rows per instruction), %s: %8.3f ms  %7.1f GB/s/CU  %6.2f TB/s\n", pat == 1 ? 64 : 128,
                   pat == 1 ? 16 : 8, hot ? "all blocks same tile (L1/L2 hot)" : "private 240 KB tile per block", ms,
                   bytes / ms / 1e6 / 256, bytes / ms / 1e9);
        }
    return 0;
}
