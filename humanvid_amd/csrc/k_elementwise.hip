// k_elementwise.hip -- translation unit for hv_elementwise.h (see hv_kernels.h)
#include "hv_elementwise.h"
#include "hv_kernels.h"

void hvk_pack(const void* src, int src_bf16, int B, int C, int Fsrc, int H, int W, const int* frames, int F, int rep,
              bf16_t* dst, int Cpad, hipStream_t s) {
    hv_pack_launch(src, src_bf16, B, C, Fsrc, H, W, frames, F, rep, dst, Cpad, s);
}
void hvk_unpack(const bf16_t* src, int ldc, int B, int C, int F, int H, int W, void* dst, int dst_bf16, hipStream_t s) {
    hv_unpack_launch(src, ldc, B, C, F, H, W, dst, dst_bf16, s);
}
void hvk_unshuffle(const float* src, int B, int C, int F, int H, int W, int r, bf16_t* dst, hipStream_t s) {
    hv_unshuffle_launch(src, B, C, F, H, W, r, dst, s);
}
void hvk_plucker(const float* K, const float* c2w, int F, int H, int W, int r, bf16_t* dst, hipStream_t s) {
    hv_plucker_launch(K, c2w, F, H, W, r, dst, s);
}
void hvk_timestep(const float* t, int B, int dim, bf16_t* dst, hipStream_t s) { hv_timestep_launch(t, B, dim, dst, s); }
void hvk_accumulate(const bf16_t* pred, int ldc, int rep, int C, int f_win, int H, int W, const int* frames, int F,
                    float* acc, float* counter, hipStream_t s) {
    hv_accumulate_launch(pred, ldc, rep, C, f_win, H, W, frames, F, acc, counter, s);
}
void hvk_cfg_ddim(float* latents, float* acc, float* counter, int rep, int C, int F, int H, int W,
                  const float* coeffs, hipStream_t s) {
    hv_cfg_ddim_launch(latents, acc, counter, rep, C, F, H, W, coeffs, s);
}
void hvk_affine_apply(const bf16_t* X, long ldx, int rows, int rows_per_image, int C, const bf16_t* X2, long ldx2, int C2,
                      const float* scale, const float* shift, int act, bf16_t* Y, long ldy, hipStream_t s) {
    hv_affine_apply_launch(X, ldx, rows, rows_per_image, C, X2, ldx2, C2, scale, shift, act, Y, ldy, s);
}
