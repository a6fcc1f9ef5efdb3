// k_conv.hip -- translation unit for hv_conv.h (see hv_kernels.h)
#include "hv_conv.h"
#include "hv_kernels.h"

int hvk_conv3x3(const hv_conv3x3_params& p, hipStream_t s) { return hv_conv3x3_launch(p, s); }
void hvk_conv_use_glds(int on) { g_hv_conv_glds = on; }
void hvk_conv_use_big(int on) { g_hv_conv_big = on; }
void hvk_conv_raster(int v) { g_hv_conv_raster = v; }
void hvk_conv_use_w4(int v) { g_hv_conv_w4 = v; }
int hvk_conv3x3_gn_parts(const hv_conv3x3_params& p) { return hv_conv3x3_gn_parts_of(p); }
#ifdef HV_C4_TRACE
extern "C" int hv_c4_trace_read(unsigned long long* host_out) {  // timing builds only (see hv_conv4.h)
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_hv_c4_trace), sizeof(unsigned long long) * 2048 * 8);
}
#endif
