// k_gemm.hip -- translation unit for hv_gemm.h (see hv_kernels.h)
#include "hv_gemm.h"
#include "hv_kernels.h"

int hvk_gemm(const hv_gemm_params& p, hipStream_t s) { return hv_gemm_launch(p, s); }
void hvk_gemm_tune(int max_grid) { g_hv_gemm_max_grid = max_grid; }
void hvk_gemm_use_glds(int on) { g_hv_gemm_glds = on; }
void hvk_gemm_use_c4(int v) { g_hv_gemm_c4 = v; }
void hvk_gemm_use_wr(int v) { g_hv_gemm_wr = v; }
void hvk_gemm_use_xs(int on) { g_hv_gemm_xs = on; }
void hvk_gemm_use_w4(int on) {
    g_hv_gemm_w4 = on == 4 ? 3 : on;
    g_hv_gemm_w4_units = on != 4;
}
int hvk_gemm_gn_parts(const hv_gemm_params& p) { return hv_gemm_gn_parts_of(p); }
int hvk_gemm_ln_parts(const hv_gemm_params& p) { return hv_gemm_ln_parts_of(p); }
#ifdef HV_W4_TRACE
extern "C" int hv_w4_trace_read(unsigned long long* host_out) {  // timing builds only (see hv_gemm4.h)
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_hv_w4_trace), sizeof(unsigned long long) * 256 * 8);
}
#endif
