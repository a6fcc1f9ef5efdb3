// Phase timeline of the LDS-DMA GEMM on one workgroup: build with
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ihumanvid_amd/csrc -DHV_GEMM_TRACE=0 tools/gemm_trace.hip -o tools/bin/gemm_trace
// Prints, per k-step: wait / barrier / issue / ds_read+MFMA durations and the epilogue time (s_memtime ticks, 100 MHz).
#include "hv_kernels.h"
#include "hv_gemm.h"
#include <cstdio>
#include <vector>

thread_local HvCmdList* g_hv_recording = nullptr;

int main(int argc, char** argv) {
    int M = 294912, N = 960, K = 320;
    if (argc > 3) M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
    const int geglu = argc > 4 ? atoi(argv[4]) : 0;
    uint16_t *X, *W, *Y;
    float* bias;
    hipMalloc(&X, (size_t)M * K * 2);
    hipMalloc(&W, (size_t)N * K * 2);
    hipMalloc(&Y, (size_t)M * N * 2);
    hipMalloc(&bias, N * 4);
    hipMemset(X, 0x3c, (size_t)M * K * 2);
    hipMemset(W, 0x3c, (size_t)N * K * 2);
    hipMemset(bias, 0, N * 4);
    HvGemmParams p{};
    p.X = X, p.ldx = K, p.W = W, p.Y = Y, p.ldy = N, p.M = M, p.N = N, p.K = K, p.bias = bias;
    if (geglu) p.geglu = 1, p.ldy = N / 2;
    p.pe_period = p.pe_frames = p.rowvec_period = p.rows_per_image = 1;
    for (int it = 0; it < 3; ++it) hv_gemm_launch(p, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(8192);
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_hv_trace), 8192 * 8);
    unsigned long long prev = 0, step0 = 0;
    double acc[16] = {0};
    int nstep = 0;
    for (int i = 0; i < 8192 && h[i]; ++i) {
        int id = h[i] >> 56;
        unsigned long long t = h[i] & 0xffffffffffffffull;
        if (id == 1) {
            if (nstep < 64 && nstep > 0) printf("\n");
            if (nstep < 64) printf("step %3d t=%8llu:", nstep, step0 ? t - step0 : 0);
            if (!step0) step0 = t;
            nstep++;
        } else {
            if (nstep <= 64) printf(" [%d]+%llu", id, t - prev);
            acc[id] += t - prev;
        }
        prev = t;
    }
    const int nt = nstep / (K / 32);
    printf("\nsteps %d; mean ticks: wait %.1f barrier %.1f issue %.1f mfma %.1f | epilogue per tile: loads->first use %.1f, 1st store %.1f, "
           "3 more stores %.1f, next fragment (4 stores) %.1f, remaining %.1f, tail %.1f; mf=2: math before each store %.1f, store instr %.1f\n",
           nstep, acc[2] / nstep, acc[3] / nstep, acc[4] / nstep, acc[5] / nstep, acc[7] / nt, acc[8] / nt, acc[9] / nt, acc[10] / nt,
           acc[11] / nt, acc[6] / nt, acc[12] / nt / (geglu ? 2 : 4), acc[13] / nt / (geglu ? 2 : 4));
    return 0;
}
