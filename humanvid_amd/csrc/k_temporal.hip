// k_temporal.hip -- translation unit for hv_temporal.h (see hv_kernels.h)
#include "hv_temporal.h"
#include "hv_kernels.h"

int hvk_temporal(const hv_temporal_attention_params& p, hipStream_t s) { return hv_temporal_launch(p, s); }
