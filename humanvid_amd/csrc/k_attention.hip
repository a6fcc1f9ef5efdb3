// k_attention.hip -- translation unit for hv_attention.h (see hv_kernels.h)
#include "hv_attention.h"
#include "hv_attention_fp8.h"
#include "hv_kernels.h"

int hvk_attention(const hv_attention_params& p, hipStream_t s) {
    return hv_attention_launch(p, s);
}
void hvk_attention_tune(int head_dim, int v) {
    if (head_dim == 40) g_hv_attn40 = v != 2;  // 0: hv_attention40, 2: the generic kernel (A/B)
}
int hvk_attention_fp8_quantize(const bf16_t* K, long ldk, const bf16_t* Vt, long ldvt, int n, int heads, int D, int L, float* kscale,
                               float* vamax, const float* vfloor, unsigned char* K8, long ldk8, unsigned char* Vt8, long ldvt8,
                               int phase, hipStream_t s) {
    return hv_attention_fp8_quantize_launch(K, ldk, Vt, ldvt, n, heads, D, L, kscale, vamax, vfloor, K8, ldk8, Vt8, ldvt8, phase, s);
}
int hvk_attention_fp8(const hv_attention_params& p, const float* ks, const float* va, const float* ks2, const float* va2,
                      hipStream_t s) {
    return hv_attention_fp8_launch(p, ks, va, ks2, va2, s);
}
