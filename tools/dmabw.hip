// LDS-DMA throughput of a CU under the GEMM's access pattern, with nothing else in the kernel (round 5).
// The timing ablations of hv_gemm_p8_kernel (profiles/r05_s2.txt) say that the k-loop of the 256 x 256 x 64 tile runs at the
// speed of its LDS-DMA alone (64 pieces of 1 KiB per k-tile and CU: ~17 B/clk/CU) whatever the schedule, while tools/fillbw
// measures 60-70 B/clk/CU for contiguous private reads.  This probe issues exactly the GEMM's stream -- 8 waves, per k-tile 4 X
// pieces + 4 W pieces per wave (8 rows x 128 B at the operand's row pitch, source-side swizzle), XCD-contiguous tile ranges with
// the grouped raster, 2-slot 64 KiB ring -- and varies one thing at a time:
//   SRC  0 GEMM sharing (X k-slice read by 4-8 CUs of an XCD, W by 8)   1 private rows per workgroup (no sharing)
//        2 private contiguous 1 KiB pieces                                3 every workgroup the same tile (L1 / L2 hot)
//   SYNC 0 vmcnt(8) + s_barrier per k-tile (one k-tile ahead)            1 free-running, vmcnt(8) only      2 vmcnt(0) + barrier
//   FORM 0 inline asm, scalar base + 32-bit lane offset, M0 saved / restored per piece   1 the builtin (64-bit lane addresses)
//        2 asm without the M0 save / restore (M0 written once per piece, clobbered)
//   WAVES 8 / 4 / 2: how many of the 8 waves issue (each then issues 8 / 16 / 32 pieces per k-tile)
// Build: hipcc --offload-arch=gfx950 -O3 tools/dmabw.hip -o tools/bin/dmabw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int SRC, int SYNC, int FORM, int WAVES>
__global__ __launch_bounds__(512, 2) void dma_kernel(const char* __restrict__ X, const char* __restrict__ W, int M, int N, int K, int gm,
                                                      unsigned* sink) {
    constexpr int XT = 32768, SLOT = 65536;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)smem;
    const int tiles_n = N / 256, tiles_m = M / 256, total = tiles_n * tiles_m;
    const int wg_per_xcd = gridDim.x / 8, xcd = blockIdx.x % 8, wg = blockIdx.x / 8;
    const int per_xcd = (total + 7) / 8, t_begin = xcd * per_xcd, t_end = min(total, t_begin + per_xcd);
    const int nk = K / 64;
    const unsigned pitch = (unsigned)K * 2u;
    const int sub = lane >> 3, prow = 8 * wave + sub;
    const unsigned chunk = (unsigned)(((lane & 7) ^ ((prow >> 1) & 7)) * 16);
    constexpr int PW = 8 / WAVES;  // pieces per operand and issuing wave multiply by this
    if (wave >= WAVES) {
        // idle waves only take part in the barriers
        for (int ti = t_begin + wg; ti < t_end; ti += wg_per_xcd)
            for (int k = 0; k < nk; ++k)
                if (SYNC != 1) __builtin_amdgcn_s_barrier();
        return;
    }
    int slot = 0;
    for (int ti = t_begin + wg; ti < t_end; ti += wg_per_xcd) {
        int m0, n0;
        {
            const int per_group = gm * tiles_n, g = ti / per_group, r = ti - g * per_group;
            const int rows = max(1, min(gm, tiles_m - g * gm));
            m0 = (g * gm + r % rows) * 256;
            n0 = (r / rows) * 256;
        }
        if (SRC == 1) m0 = (blockIdx.x % (M / 256)) * 256, n0 = (blockIdx.x % (N / 256)) * 256;
        if (SRC == 3) m0 = 0, n0 = 0;
        for (int k = 0; k < nk; ++k) {
            const int kk = SRC == 3 ? 0 : k;
#pragma unroll
            for (int op = 0; op < 2; ++op) {
#pragma unroll
                for (int q = 0; q < 4 * PW; ++q) {
                    const int piece = (q % 4), wv = wave + WAVES * (q / 4);  // (wave, piece) of the 8-wave layout this issue stands for
                    const char* base = op == 0 ? X : W;
                    const unsigned row0 = (unsigned)((op == 0 ? m0 : n0) + 64 * piece);
                    unsigned lane_ofs, tile_ofs;
                    if (SRC == 2) {  // contiguous 1 KiB pieces from a private 64 KiB-per-k-tile region
                        tile_ofs = ((unsigned)blockIdx.x * (unsigned)nk + (unsigned)kk) * 65536u % (1u << 30) + (unsigned)(op * 32768 + (piece * 8 + wv) * 1024);
                        lane_ofs = (unsigned)lane * 16u;
                    } else {
                        tile_ofs = row0 * pitch + (unsigned)kk * 128u;
                        lane_ofs = (unsigned)(8 * wv + sub) * pitch + (unsigned)(((lane & 7) ^ (((8 * wv + sub) >> 1) & 7)) * 16);
                        (void)chunk;
                    }
                    const unsigned lds = lds0 + slot * SLOT + op * XT + (wv + 8 * piece) * 1024;
                    if (FORM == 1) {
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + tile_ofs + lane_ofs),
                                                         (__attribute__((address_space(3))) void*)(smem + slot * SLOT + op * XT + (wv + 8 * piece) * 1024), 16, 0, 0);
                    } else if (FORM == 0) {
                        unsigned keep;
                        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                                     : "=&s"(keep)
                                     : "v"(lane_ofs), "s"(base + tile_ofs), "s"(lds)
                                     : "memory");
                    } else {
                        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_ofs), "s"(base + tile_ofs), "s"(lds)
                                     : "memory", "m0");
                    }
                }
            }
            slot ^= 1;
            if (SYNC == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * PW) : "memory");
            if (SYNC != 1) __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (reinterpret_cast<unsigned*>(smem)[tid] == 0x12345) sink[0] = 1;
}

template <int SRC, int SYNC, int FORM, int WAVES>
static void run(const char* name, const char* X, const char* W, int M, int N, int K, unsigned* sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int gm = N / 128 > 8 ? 8 : 1;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((dma_kernel<SRC, SYNC, FORM, WAVES>), dim3(256), dim3(512), 0, 0, X, W, M, N, K, gm, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)(M / 256) * (N / 256) * (K / 64) * 65536.0;
    printf("%-78s M=%6d N=%5d K=%4d  %7.3f ms  %6.1f GB/s/CU  %5.1f B/clk/CU @2.1GHz  %6.2f TB/s\n", name, M, N, K, best,
           bytes / best / 1e6 / 256, bytes / best / 1e6 / 256 / 2.1, bytes / best / 1e9);
}

int main() {
    char *X, *W;
    unsigned* sink;
    hipMalloc(&X, 1u << 30);
    hipMalloc(&W, 1u << 30);
    hipMemset(X, 1, 1u << 30);
    hipMemset(W, 1, 1u << 30);
    hipMalloc(&sink, 4);
    struct Shape { int M, N, K; } shapes[] = {{18432, 10240, 1280}, {73728, 5120, 640}, {294912, 2560, 320}};
    for (auto& s : shapes) {
        run<0, 0, 0, 8>("GEMM stream: asm, vmcnt(8) + barrier per k-tile", X, W, s.M, s.N, s.K, sink);
        run<0, 1, 0, 8>("  free-running (no barrier)", X, W, s.M, s.N, s.K, sink);
        run<0, 2, 0, 8>("  vmcnt(0) + barrier per k-tile (nothing in flight across it)", X, W, s.M, s.N, s.K, sink);
        run<0, 0, 1, 8>("  builtin global_load_lds (64-bit lane addresses)", X, W, s.M, s.N, s.K, sink);
        run<0, 0, 2, 8>("  asm, M0 not saved / restored", X, W, s.M, s.N, s.K, sink);
        run<0, 0, 0, 4>("  4 of 8 waves issue (16 pieces each)", X, W, s.M, s.N, s.K, sink);
        run<0, 0, 0, 2>("  2 of 8 waves issue (32 pieces each)", X, W, s.M, s.N, s.K, sink);
        run<1, 0, 0, 8>("  private rows per workgroup (no sharing between CUs), L2-resident", X, W, s.M, s.N, s.K, sink);
        run<2, 0, 0, 8>("  private contiguous 1 KiB pieces (streaming)", X, W, s.M, s.N, s.K, sink);
        run<3, 0, 0, 8>("  every workgroup the same k-tile (L1 / L2 hot)", X, W, s.M, s.N, s.K, sink);
        run<3, 1, 0, 8>("  ... hot and free-running", X, W, s.M, s.N, s.K, sink);
    }
    return 0;
}
