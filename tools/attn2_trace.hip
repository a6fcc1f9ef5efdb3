// Phase timeline of the round-2 level-0 spatial attention (d = 40, 6144 queries, 6144 own keys): wave 0 of one workgroup
// logs s_memtime per pipeline step at: 1 step start, 2 after the counted vmcnt wait, 3 after the barrier, 4 after issuing
// the DMA group, 5 after max / rescale, 6 after region 1 (S^T of the next tile || probabilities of block 0, V^T reads
// waited), 7 after region 2 (P.V block 0 || probabilities block 1, P.V block 1).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ihumanvid_amd/csrc -DHV_GEMM_TRACE=<workgroup> tools/attn2_trace.hip -o tools/bin/attn2_trace
#include "hv_kernels.h"
#include "hv_gemm.h"
#include "hv_attention2.h"
#include <cstdio>
#include <vector>
thread_local HvCmdList* g_hv_recording = nullptr;
thread_local HvProfile* g_hv_prof = nullptr;
thread_local char g_hv_note[192] = "";

int main() {
    const int n_img = 48, heads = 8, D = 40, N = 6144, C = 320;
    const long M = (long)n_img * N;
    uint16_t *qkv, *o;
    hipMalloc(&qkv, M * 3 * C * 2);
    hipMalloc(&o, M * C * 2);
    hipMemset(qkv, 0x3c, M * 3 * C * 2);
    hv_attention_params p{};
    p.Q = qkv, p.K = qkv + C, p.Vt = qkv + 2 * C, p.O = o;
    p.ldq = 3 * C, p.ldk = 3 * C, p.ldvt = 3 * C, p.ldo = C;
    p.n_images = n_img, p.heads = heads, p.D = D, p.Lq = N, p.L1 = N, p.scale = 0.158f, p.v_row_major = 1;
    hv_attention2_launch(p, 0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hv_attention2_launch(p, 0);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("kernel %.3f ms (48 images x 6144 queries x 6144 own keys, no bank; HV_A2_DBG=%d)\n", ms, HV_A2_DBG);
    std::vector<unsigned long long> h(8192);
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_hv_trace), 8192 * 8);
    double acc[8] = {0};
    unsigned long long prev = 0, t0 = 0;
    int n = 0;
    for (int i = 0; i < 8192 && h[i]; ++i) {
        const int id = h[i] >> 56;
        const unsigned long long t = h[i] & 0xffffffffffffffull;
        if (id == 1) {
            if (n < 10) printf("%sstep %2d t=%7llu:", n ? "\n" : "", n, t0 ? t - t0 : 0);
            if (!t0) t0 = t;
            if (n) acc[1] += t - prev;
            ++n;
        } else {
            if (n <= 10) printf(" [%d]+%llu", id, t - prev);
            acc[id] += t - prev;
        }
        prev = t;
    }
    printf("\nsteps %d; mean cycles: vmcnt wait %.0f, barrier %.0f, DMA issue %.0f, max/rescale %.0f, region 1 (S next || probs 0) %.0f, "
           "region 2 (PV || probs 1) %.0f, loop-back %.0f\n", n, acc[2] / n, acc[3] / n, acc[4] / n, acc[5] / n, acc[6] / n, acc[7] / n,
           acc[1] / n);
    return 0;
}
