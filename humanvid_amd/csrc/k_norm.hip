// k_norm.hip -- translation unit for hv_norm.h (see hv_kernels.h)
#include "hv_norm.h"
#include "hv_kernels.h"

int hvk_groupnorm(const hv_groupnorm_params& p, hipStream_t s) { return hv_groupnorm_launch(p, s); }
void hvk_layernorm(const bf16_t* X, long ldx, int M, int C, float eps, float* mean, float* rstd, hipStream_t s) {
    hv_layernorm_launch(X, ldx, M, C, eps, mean, rstd, s);
}
int hvk_gn_from_parts(const hv_gn_parts_params& p, hipStream_t s) { return hv_gn_from_parts_launch(p, s); }
int hvk_ln_from_parts(const float* part, int parts, int M, int C, float eps, float* mean, float* rstd, hipStream_t s) {
    return hv_ln_from_parts_launch(part, parts, M, C, eps, mean, rstd, s);
}
