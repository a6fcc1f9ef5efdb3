// Phase timeline of the LDS-DMA GEMM on one workgroup (wave 0 of workgroup HV_GEMM_TRACE logs s_memtime at phase marks):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ihumanvid_amd/csrc -DHV_GEMM_TRACE=0 tools/gemm_trace.hip -o tools/bin/gemm_trace
//   tools/bin/gemm_trace M N K form tile_policy        form: 0 plain, 1 LN fold, 2 LN + GEGLU, 3 residual (in place)
// Marks: 1 k-step top, 2 after the vmcnt wait, 3 after the barrier, 4 after the LDS-DMA issue, 5 after ds_read + MFMA,
// 7 epilogue loads of the first group landed, 12 packed outputs ready (before the stores), 13 stores issued, 6 tile done.
#include "hv_kernels.h"
#include "hv_gemm.h"
#include <cstdio>
#include <vector>

thread_local HvCmdList* g_hv_recording = nullptr;
thread_local HvProfile* g_hv_prof = nullptr;
thread_local char g_hv_note[192] = "";

int main(int argc, char** argv) {
    int M = 294912, N = 960, K = 320, form = 1, policy = 1;
    if (argc > 3) M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
    if (argc > 4) form = atoi(argv[4]);
    if (argc > 5) policy = atoi(argv[5]);
    g_hv_gemm_glds = policy;
    const int geglu = form == 2;
    uint16_t *X, *W, *Y;
    float *bias, *mean, *rstd, *cs;
    hipMalloc(&X, (size_t)M * K * 2);
    hipMalloc(&W, (size_t)N * K * 2);
    hipMalloc(&Y, (size_t)M * N * 2);
    hipMalloc(&bias, N * 4);
    hipMalloc(&cs, N * 4);
    hipMalloc(&mean, (size_t)M * 4);
    hipMalloc(&rstd, (size_t)M * 4);
    hipMemset(X, 0x3c, (size_t)M * K * 2);
    hipMemset(W, 0x3c, (size_t)N * K * 2);
    hipMemset(Y, 0, (size_t)M * N * 2);
    hipMemset(bias, 0, N * 4);
    hipMemset(cs, 0, N * 4);
    hipMemset(mean, 0, (size_t)M * 4);
    hipMemset(rstd, 0, (size_t)M * 4);
    HvGemmParams p{};
    p.X = X, p.ldx = K, p.W = W, p.Y = Y, p.ldy = N, p.M = M, p.N = N, p.K = K, p.bias = bias;
    if (form == 1 || form == 2) p.row_mean = mean, p.row_rstd = rstd, p.colsum = cs;
    if (geglu) p.geglu = 1, p.ldy = N / 2;
    if (form == 3) p.residual = Y, p.ldr = N;
    p.pe_period = p.pe_frames = p.rowvec_period = p.rows_per_image = 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hv_gemm_launch(p, 0);
    hipEventRecord(e0, 0);
    for (int it = 0; it < 3; ++it) hv_gemm_launch(p, 0);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(8192);
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_hv_trace), 8192 * 8);
    unsigned long long prev = 0, t_first = 0, t_last = 0;
    double acc[16] = {0};
    int cnt[16] = {0};
    int prev_id = 0, ntile = 0, nstep = 0;
    for (int i = 0; i < 8192 && h[i]; ++i) {
        const int id = h[i] >> 56;
        const unsigned long long t = h[i] & 0xffffffffffffffull;
        if (!t_first) t_first = t;
        t_last = t;
        if (i > 0) {
            // attribute the interval to the mark that ENDS it, keyed by (prev_id -> id) for the epilogue pieces
            acc[id] += (double)(t - prev), cnt[id]++;
        }
        if (id == 1) nstep++;
        if (id == 6) ntile++;
        prev = t, prev_id = id;
        if (i < 48) printf("%s[%d]%llu", i ? " " : "", id, i ? t - (h[i - 1] & 0xffffffffffffffull) : 0ull);
    }
    (void)prev_id;
    printf("\n");
    auto avg = [&](int id) { return cnt[id] ? acc[id] / cnt[id] : 0.0; };
    printf("M=%d N=%d K=%d form=%d policy=%d: %.3f ms/launch; workgroup %d logged %d k-steps, %d tiles over %llu ticks (%.0f per tile)\n", M, N, K,
           form, policy, ms / 3, HV_GEMM_TRACE, nstep, ntile, t_last - t_first, ntile ? (double)(t_last - t_first) / ntile : 0.0);
    printf("  per k-step: vmcnt wait %.0f, barrier %.0f, DMA issue %.0f, ds_read+MFMA %.0f, loop bookkeeping %.0f\n", avg(2), avg(3),
           avg(4), avg(5), avg(1));
    printf("  per tile epilogue: first loads landed %.0f, -> outputs packed %.0f, store issue %.0f, end %.0f, clear %.0f\n", avg(7),
           avg(12), avg(13), avg(11), avg(6));
    printf("  GEGLU epilogue stores: vmcnt(0) in front of them %.0f (= write-acknowledge latency of the trace's own mark stores), first store %.0f, stores 2-4 %.0f, stores 5-8 %.0f\n", avg(9), avg(8),
           avg(10), avg(13));
    return 0;
}
