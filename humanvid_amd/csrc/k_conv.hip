// k_conv.hip -- translation unit for hv_conv.h (see hv_kernels.h)
#include "hv_conv.h"
#include "hv_kernels.h"

int hvk_conv3x3(const hv_conv3x3_params& p, hipStream_t s) { return hv_conv3x3_launch(p, s); }
