// hv_kernels.h -- launch entry points of the kernel translation units (k_*.hip).
// Each family is its own TU so hipcc can build them in parallel; the emulator build (tests)
// defines HV_SINGLE_TU and pulls the kernels in here instead.
#pragma once
#include "hv_common.h"

int hvk_gemm(const hv_gemm_params& p, hipStream_t s);
int hvk_gemm_gn_parts(const hv_gemm_params& p);
int hvk_gemm_ln_parts(const hv_gemm_params& p);
void hvk_gemm_tune(int max_grid);
void hvk_gemm_use_glds(int on);
void hvk_gemm_use_w4(int on);
void hvk_gemm_use_c4(int v);
void hvk_gemm_use_wr(int v);
void hvk_gemm_use_xs(int on);
void hvk_conv_use_glds(int on);
void hvk_conv_use_big(int on);
void hvk_conv_raster(int v);
void hvk_conv_use_w4(int v);
int hvk_conv3x3(const hv_conv3x3_params& p, hipStream_t s);
int hvk_conv3x3_gn_parts(const hv_conv3x3_params& p);
int hvk_groupnorm(const hv_groupnorm_params& p, hipStream_t s);
int hvk_gn_from_parts(const hv_gn_parts_params& p, hipStream_t s);
int hvk_ln_from_parts(const float* part, int parts, int M, int C, float eps, float* mean, float* rstd, hipStream_t s);
void hvk_layernorm(const bf16_t* X, long ldx, int M, int C, float eps, float* mean, float* rstd, hipStream_t s);
int hvk_attention(const hv_attention_params& p, hipStream_t s);
void hvk_attention_tune(int head_dim, int qt);
int hvk_attention_fp8_quantize(const bf16_t* K, long ldk, const bf16_t* Vt, long ldvt, int n, int heads, int D, int L, float* kscale,
                               float* vamax, const float* vfloor, unsigned char* K8, long ldk8, unsigned char* Vt8, long ldvt8,
                               int phase, hipStream_t s);
int hvk_attention_fp8(const hv_attention_params& p, const float* ks, const float* va, const float* ks2, const float* va2,
                      hipStream_t s);
int hvk_temporal(const hv_temporal_attention_params& p, hipStream_t s);
void hvk_pack(const void* src, int src_bf16, int B, int C, int Fsrc, int H, int W, const int* frames, int F, int rep,
              bf16_t* dst, int Cpad, hipStream_t s);
void hvk_unpack(const bf16_t* src, int ldc, int B, int C, int F, int H, int W, void* dst, int dst_bf16, hipStream_t s);
void hvk_unshuffle(const float* src, int B, int C, int F, int H, int W, int r, bf16_t* dst, hipStream_t s);
void hvk_plucker(const float* K, const float* c2w, int F, int H, int W, int r, bf16_t* dst, hipStream_t s);
void hvk_timestep(const float* t, int B, int dim, bf16_t* dst, hipStream_t s);
void hvk_accumulate(const bf16_t* pred, int ldc, int rep, int C, int f_win, int H, int W, const int* frames, int F,
                    float* acc, float* counter, hipStream_t s);
void hvk_cfg_ddim(float* latents, float* acc, float* counter, int rep, int C, int F, int H, int W,
                  const float* coeffs, hipStream_t s);
void hvk_affine_apply(const bf16_t* X, long ldx, int rows, int rows_per_image, int C, const bf16_t* X2, long ldx2, int C2,
                      const float* scale, const float* shift, int act, bf16_t* Y, long ldy, hipStream_t s);

#ifdef HV_SINGLE_TU
#include "k_gemm.hip"
#include "k_conv.hip"
#include "k_norm.hip"
#include "k_attention.hip"
#include "k_temporal.hip"
#include "k_elementwise.hip"
#endif
