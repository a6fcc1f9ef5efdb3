// Where the waves of one hv_attention40 workgroup spend their cycles (scalar clock sums per wave: staging, MFMA phases, VALU
// phases, barriers):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ihumanvid_amd/csrc -DHV_ATTN40_TRACE=0 tools/attn40_trace.hip -o tools/bin/attn40_trace
//   tools/bin/attn40_trace [n_images] [L]
#include "hv_kernels.h"
#include "hv_attention.h"
#include <cstdio>
#include <vector>

thread_local HvCmdList* g_hv_recording = nullptr;
thread_local HvProfile* g_hv_prof = nullptr;
thread_local char g_hv_note[192] = "";

int main(int argc, char** argv) {
    int n = 48, L = 6144;
    if (argc > 1) n = atoi(argv[1]);
    if (argc > 2) L = atoi(argv[2]);
    const int C = 320, heads = 8;
    const size_t M = (size_t)n * L;
    uint16_t *qk, *vt, *k2, *vt2, *o;
    int* sel;
    hipMalloc(&qk, M * 2 * C * 2);
    hipMalloc(&vt, (size_t)C * M * 2);
    hipMalloc(&k2, (size_t)2 * L * C * 2);
    hipMalloc(&vt2, (size_t)C * 2 * L * 2);
    hipMalloc(&o, M * C * 2);
    hipMalloc(&sel, n * 4);
    hipMemset(qk, 0x3c, M * 2 * C * 2);   // bf16 0x3c3c = 0.0115
    hipMemset(vt, 0x3c, (size_t)C * M * 2);
    hipMemset(k2, 0x3c, (size_t)2 * L * C * 2);
    hipMemset(vt2, 0x3c, (size_t)C * 2 * L * 2);
    std::vector<int> hs(n);
    for (int i = 0; i < n; ++i) hs[i] = i < n / 2 ? -1 : 1;
    hipMemcpy(sel, hs.data(), n * 4, hipMemcpyHostToDevice);
    hv_attention_params p{};
    p.Q = qk, p.ldq = 2 * C, p.K = qk + C, p.ldk = 2 * C, p.Vt = vt, p.ldvt = (long)M, p.O = o, p.ldo = C;
    p.K2 = k2, p.ldk2 = C, p.Vt2 = vt2, p.ldvt2 = 2L * L, p.L2 = L, p.bank_sel = sel;
    p.n_images = n, p.heads = heads, p.D = 40, p.Lq = L, p.L1 = L, p.scale = 0.158f;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hv_attention40_launch(p, 0);
    hipEventRecord(e0, 0);
    for (int it = 0; it < 3; ++it) hv_attention40_launch(p, 0);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(64);
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_hv_a40_trace), 64 * 8);
    printf("hv_attention40 n=%d L=%d: %.3f ms/launch; workgroup %d:\n", n, L, ms / 3, HV_ATTN40_TRACE);
    for (int w = 0; w < 8; ++w) {
        const unsigned long long* r = h.data() + w * 8;
        printf("  wave %d: %llu tiles, loop %llu cycles = %.0f per tile: staging %.0f, M phase %.0f (x%llu), V phase %.0f (x%llu), barriers %.0f per tile\n", w,
               r[7], r[0], (double)r[0] / r[7], (double)r[1] / r[7], r[5] ? (double)r[2] / r[5] : 0.0, r[5], r[6] ? (double)r[3] / r[6] : 0.0,
               r[6], (double)r[4] / r[7]);
    }
    return 0;
}
