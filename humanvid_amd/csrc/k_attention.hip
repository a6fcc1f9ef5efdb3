// k_attention.hip -- translation unit for hv_attention.h (see hv_kernels.h)
#include "hv_attention.h"
#include "hv_kernels.h"

int hvk_attention(const hv_attention_params& p, hipStream_t s) { return hv_attention_launch(p, s); }
