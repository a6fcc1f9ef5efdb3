// k_conv.hip -- translation unit for hv_conv.h (see hv_kernels.h)
#include "hv_conv.h"
#include "hv_kernels.h"

int hvk_conv3x3(const hv_conv3x3_params& p, hipStream_t s) { return hv_conv3x3_launch(p, s); }
void hvk_conv_use_glds(int on) { g_hv_conv_glds = on; }
void hvk_conv_use_big(int on) { g_hv_conv_big = on; }
void hvk_conv_raster(int v) { g_hv_conv_raster = v; }
int hvk_conv3x3_gn_parts(const hv_conv3x3_params& p) { return hv_conv3x3_gn_parts_of(p); }
