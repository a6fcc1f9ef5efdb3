// Phase timeline of the level-0 spatial attention (d = 40, 6144 queries, 6144 keys): wave 0 of one workgroup logs
// s_memtime at: 1 tile start, 2 after the LDS tile store, 3 after the barrier, 4 after issuing the next tile's global
// loads, 5 after Q.K^T, 6 after the softmax, 7 after P.V.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ihumanvid_amd/csrc -DHV_GEMM_TRACE=<workgroup> tools/attn_trace.hip -o tools/bin/attn_trace
#include "hv_kernels.h"
#include "hv_gemm.h"
#include "hv_attention.h"
#include <cstdio>
#include <vector>
thread_local HvCmdList* g_hv_recording = nullptr;
thread_local HvProfile* g_hv_prof = nullptr;
thread_local char g_hv_note[192];

int main() {
    const int n_img = 48, heads = 8, D = 40, N = 6144, C = 320;
    const long M = (long)n_img * N;
    uint16_t *qk, *vt, *o;
    hipMalloc(&qk, M * 2 * C * 2);
    hipMalloc(&vt, (size_t)C * M * 2);
    hipMalloc(&o, M * C * 2);
    hipMemset(qk, 0x3c, M * 2 * C * 2);
    hipMemset(vt, 0x3c, (size_t)C * M * 2);
    hv_attention_params p{};
    p.Q = qk, p.K = qk + C, p.Vt = vt, p.O = o;
    p.ldq = 2 * C, p.ldk = 2 * C, p.ldvt = M, p.ldo = C;
    p.n_images = n_img, p.heads = heads, p.D = D, p.Lq = N, p.L1 = N, p.scale = 0.158f;
    for (int it = 0; it < 2; ++it) hv_attention_launch(p, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(8192);
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_hv_trace), 8192 * 8);
    double acc[8] = {0};
    unsigned long long prev = 0, t0 = 0;
    int n = 0;
    for (int i = 0; i < 8192 && h[i]; ++i) {
        const int id = h[i] >> 56;
        const unsigned long long t = h[i] & 0xffffffffffffffull;
        if (id == 1) {
            if (n < 12) printf("%stile %2d t=%7llu:", n ? "\n" : "", n, t0 ? t - t0 : 0);
            if (!t0) t0 = t;
            if (n) acc[1] += t - prev;
            ++n;
        } else {
            if (n <= 12) printf(" [%d]+%llu", id, t - prev);
            acc[id] += t - prev;
        }
        prev = t;
    }
    printf("\ntiles %d; mean cycles: LDS store %.0f, barrier %.0f, global-load issue %.0f, QK^T %.0f, softmax %.0f, PV %.0f, loop-back %.0f\n", n,
           acc[2] / n, acc[3] / n, acc[4] / n, acc[5] / n, acc[6] / n, acc[7] / n, acc[1] / n);
    return 0;
}
