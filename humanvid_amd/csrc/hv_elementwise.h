// hv_elementwise.h -- layout adaptors at the drop-in boundary and the per-step scalar-rate tail
// (window accumulation, classifier-free guidance, DDIM v-prediction update).  All HBM-bound,
// grid-stride, one thread per output vector.
#pragma once
#include "hv_common.h"
#include "humanvid_hip.h"

// [b][c][f][h][w] (fp32 / bf16)  ->  [(rep b) f][h][w][Cpad] bf16, zero channel padding.
// The reference's 'b c f h w -> (b f) c h w' rearrange (src/models/resnet.py:12) plus the CFG
// `.repeat(2, ...)` of src/pipelines/pipeline_pose2vid_long.py:516-520 in one pass.
__global__ __launch_bounds__(256) void hv_pack_kernel(const void* src, int src_bf16, int B, int C, int Fsrc, int H,
                                                      int W, const int* frames, int F, int rep, bf16_t* dst,
                                                      int Cpad) {
    const long npix = (long)B * F * H * W;
    const int cvs = Cpad / 8;
    const long total = npix * cvs;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % cvs);
        long pix = i / cvs;
        const int x = (int)(pix % W);
        pix /= W;
        const int y = (int)(pix % H);
        pix /= H;
        const int f = (int)(pix % F);
        const int b = (int)(pix / F);
        const int fs = frames ? frames[f] : f;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = cv * 8 + e;
            float a = 0.f;
            if (c < C) {
                const long si = ((((long)b * C + c) * Fsrc + fs) * H + y) * W + x;
                a = src_bf16 ? hv_bf2f(reinterpret_cast<const bf16_t*>(src)[si]) : reinterpret_cast<const float*>(src)[si];
            }
            v[e] = a;
        }
        const u32x4 o = hv_pack8(v);
        for (int r = 0; r < rep; ++r) {
            const long di = ((((long)(r * B + b) * F + f) * H + y) * W + x) * Cpad + cv * 8;
            hv_st16(dst + di, o);
        }
    }
}

// [(b f)][h][w][ldc] bf16 -> [b][C][f][h][w] fp32 / bf16 ('(b f) c h w -> b c f h w')
__global__ __launch_bounds__(256) void hv_unpack_kernel(const bf16_t* src, int ldc, int B, int C, int F, int H, int W,
                                                        void* dst, int dst_bf16) {
    const long total = (long)B * C * F * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i;
        const int x = (int)(r % W);
        r /= W;
        const int y = (int)(r % H);
        r /= H;
        const int f = (int)(r % F);
        r /= F;
        const int c = (int)(r % C);
        const int b = (int)(r / C);
        const bf16_t v = src[((((long)b * F + f) * H + y) * W + x) * ldc + c];
        if (dst_bf16)
            reinterpret_cast<bf16_t*>(dst)[i] = v;
        else
            reinterpret_cast<float*>(dst)[i] = hv_bf2f(v);
    }
}

// nn.PixelUnshuffle(r): out[(b f)][y][x][c*r*r + i*r + j] = in[b][c][f][y*r+i][x*r+j]
__global__ __launch_bounds__(256) void hv_unshuffle_kernel(const float* src, int B, int C, int F, int H, int W, int r,
                                                           bf16_t* dst) {
    const int Ho = H / r, Wo = W / r, Co = C * r * r;
    const long total = (long)B * F * Ho * Wo * Co;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long t = i;
        const int co = (int)(t % Co);
        t /= Co;
        const int x = (int)(t % Wo);
        t /= Wo;
        const int y = (int)(t % Ho);
        t /= Ho;
        const int f = (int)(t % F);
        const int b = (int)(t / F);
        const int c = co / (r * r), ij = co % (r * r), ii = ij / r, jj = ij % r;
        dst[i] = hv_f2bf(src[((((long)b * C + c) * F + f) * H + y * r + ii) * W + x * r + jj]);
    }
}

// ray_condition (src/dataset/dance_image_h_v_camera.py:88-130) fused with nn.PixelUnshuffle(r)
// (src/cameractrl/pose_adaptor.py:177,236): the Pluecker map (o x d, d) of frame f at pixel (Y, X) is generated from
// the frame's intrinsics K[f] = (fx, fy, cx, cy) and camera-to-world matrix c2w[f] (row-major 4x4) and written
// straight into the camera encoder's input layout out[f][Y/r][X/r][c*r*r + (Y%r)*r + (X%r)], c = 0..5 -- the
// [1,6,F,H,W] fp32 map (57 MB at 24 x 768 x 512) is never materialised.  fp32 arithmetic, bf16 store.
__global__ __launch_bounds__(256) void hv_plucker_kernel(const float* K, const float* c2w, int F, int H, int W, int r,
                                                         bf16_t* dst) {
    const int Ho = H / r, Wo = W / r, Co = 6 * r * r;
    const long total = (long)F * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int X = (int)(i % W);
        const int Y = (int)((i / W) % H);
        const int f = (int)(i / ((long)W * H));
        const float* k = K + 4 * f;
        const float* m = c2w + 16 * f;
        const float xs = ((float)X + 0.5f - k[2]) / k[0], ys = ((float)Y + 0.5f - k[3]) / k[1];
        const float inv = 1.0f / sqrtf(xs * xs + ys * ys + 1.0f);
        const float dx = xs * inv, dy = ys * inv, dz = inv;
        float d[3], o[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            d[a] = dx * m[4 * a + 0] + dy * m[4 * a + 1] + dz * m[4 * a + 2];  // directions @ R^T
            o[a] = m[4 * a + 3];
        }
        const float pl[6] = {o[1] * d[2] - o[2] * d[1], o[2] * d[0] - o[0] * d[2], o[0] * d[1] - o[1] * d[0], d[0], d[1], d[2]};
        bf16_t* out = dst + (((long)f * Ho + Y / r) * Wo + X / r) * Co + (Y % r) * r + (X % r);
#pragma unroll
        for (int c = 0; c < 6; ++c) out[c * r * r] = hv_f2bf(pl[c]);
    }
}

// diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]
__global__ __launch_bounds__(256) void hv_timestep_kernel(const float* t, int B, int dim, bf16_t* dst) {
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, k = i % half;
    const float freq = expf(-9.210340371976184f * (float)k / (float)half);  // ln(10000)
    const float a = t[b] * freq;
    dst[(long)b * dim + k] = hv_f2bf(cosf(a));
    dst[(long)b * dim + half + k] = hv_f2bf(sinf(a));
}

// noise_pred[:, :, c] += pred ; counter[:, :, c] += 1   (pipeline_pose2vid_long.py:550-552)
__global__ __launch_bounds__(256) void hv_accumulate_kernel(const bf16_t* pred, int ldc, int rep, int C, int f_win,
                                                            int H, int W, const int* frames, int F, float* acc,
                                                            float* counter) {
    const long total = (long)rep * C * f_win * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long t = i;
        const int x = (int)(t % W);
        t /= W;
        const int y = (int)(t % H);
        t /= H;
        const int fw = (int)(t % f_win);
        t /= f_win;
        const int c = (int)(t % C);
        const int r = (int)(t / C);
        const int f = frames[fw];
        const float v = hv_bf2f(pred[((((long)r * f_win + fw) * H + y) * W + x) * ldc + c]);
        acc[((((long)r * C + c) * F + f) * H + y) * W + x] += v;
        if (r == 0 && c == 0 && y == 0 && x == 0) counter[f] += 1.0f;
    }
}

// (noise_pred / counter).chunk(2) -> u + s (t - u) -> DDIM v-prediction step, eta = 0
// (pipeline_pose2vid_long.py:555-563; diffusers DDIMScheduler.step).
// Also clears the accumulators for the next step.
__global__ __launch_bounds__(256) void hv_cfg_ddim_kernel(float* latents, float* acc, float* counter, int rep, int C,
                                                          int F, int H, int W, const float* coeffs) {
    const float guidance = coeffs[0], sqrt_a = coeffs[1], sqrt_1ma = coeffs[2], sqrt_ap = coeffs[3],
                sqrt_1map = coeffs[4];
    const long per = (long)C * F * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long)gridDim.x * blockDim.x) {
        const int f = (int)((i / ((long)H * W)) % F);
        const float cnt = counter[f];
        float v = acc[i] / cnt;
        acc[i] = 0.f;
        if (rep == 2) {
            const float c = acc[per + i] / cnt;
            acc[per + i] = 0.f;
            v = v + guidance * (c - v);
        }
        const float x = latents[i];
        const float x0 = sqrt_a * x - sqrt_1ma * v;
        const float eps = sqrt_a * v + sqrt_1ma * x;
        latents[i] = sqrt_ap * x0 + sqrt_1map * eps;
    }
}

__global__ void hv_clear_kernel(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

static inline int hv_ew_grid(long total) {
    long g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

static inline void hv_pack_launch(const void* src, int src_bf16, int B, int C, int Fsrc, int H, int W,
                                  const int* frames, int F, int rep, bf16_t* dst, int Cpad, hipStream_t s) {
    hv_launch(hv_pack_kernel, dim3(hv_ew_grid((long)B * F * H * W * (Cpad / 8))), dim3(256), s, src, src_bf16, B, C,
              Fsrc, H, W, frames, F, rep, dst, Cpad);
}
static inline void hv_unpack_launch(const bf16_t* src, int ldc, int B, int C, int F, int H, int W, void* dst,
                                    int dst_bf16, hipStream_t s) {
    hv_launch(hv_unpack_kernel, dim3(hv_ew_grid((long)B * C * F * H * W)), dim3(256), s, src, ldc, B, C, F, H, W, dst,
              dst_bf16);
}
static inline void hv_unshuffle_launch(const float* src, int B, int C, int F, int H, int W, int r, bf16_t* dst,
                                       hipStream_t s) {
    hv_launch(hv_unshuffle_kernel, dim3(hv_ew_grid((long)B * C * F * H * W)), dim3(256), s, src, B, C, F, H, W, r, dst);
}
static inline void hv_plucker_launch(const float* K, const float* c2w, int F, int H, int W, int r, bf16_t* dst, hipStream_t s) {
    hv_launch(hv_plucker_kernel, dim3(hv_ew_grid((long)F * H * W)), dim3(256), s, K, c2w, F, H, W, r, dst);
}
static inline void hv_timestep_launch(const float* t, int B, int dim, bf16_t* dst, hipStream_t s) {
    hv_launch(hv_timestep_kernel, dim3((B * dim / 2 + 255) / 256), dim3(256), s, t, B, dim, dst);
}
static inline void hv_accumulate_launch(const bf16_t* pred, int ldc, int rep, int C, int f_win, int H, int W,
                                        const int* frames, int F, float* acc, float* counter, hipStream_t s) {
    hv_launch(hv_accumulate_kernel, dim3(hv_ew_grid((long)rep * C * f_win * H * W)), dim3(256), s, pred, ldc, rep, C,
              f_win, H, W, frames, F, acc, counter);
}
static inline void hv_cfg_ddim_launch(float* latents, float* acc, float* counter, int rep, int C, int F, int H, int W,
                                      const float* coeffs, hipStream_t s) {
    hv_launch(hv_cfg_ddim_kernel, dim3(hv_ew_grid((long)C * F * H * W)), dim3(256), s, latents, acc, counter, rep, C, F,
              H, W, coeffs);
    hv_launch(hv_clear_kernel, dim3((F + 255) / 256), dim3(256), s, counter, F);
}

// y[(img row)][c] = act(x * scale[img][c] + shift[img][c]) -- GroupNorm apply as its own pass (bf16 -> bf16).
// Reference: the InflatedGroupNorm in front of Transformer3DModel.proj_in / TemporalTransformer3DModel.proj_in
// (src/models/transformer_3d.py:125-131, src/models/motion_module.py:157-163).  Round 1 applied it as a prologue on the
// GEMM's A operand, which forces the register-staged GEMM kernel (8.5 % MFMA-busy: unpack / fma / pack per 16-byte chunk
// in front of every LDS store); as a separate HBM-bound pass it costs 4 bytes per element and the projection runs on the
// LDS-DMA kernel.  One thread per 8 channels, rows grid-strided; scale / shift rows are L1/L2 hits.
// Two sources (X2 != nullptr): the channel concatenation [X | X2] of a decoder ResnetBlock3D's input
// (unet_3d_blocks.py: torch.cat([hidden_states, res_hidden_states], dim=1) in front of resnet.py:215-245) is written as ONE
// [rows, C + C2] activation, so the convolution behind it reads a single source; scale / shift rows are C + C2 wide.
// Round 4 form: a thread keeps ONE 8-channel slot of ONE image (its scale / shift pairs stay in 16 registers) and walks rows of
// that image, four independent 16-byte loads in flight -- the round-3 form re-derived (row, slot, image) with 64-bit
// divisions and re-loaded 64 bytes of scale / shift for every 16 bytes of payload (five VMEM instructions per store): it ran
// at 4.1 TB/s where a copy reaches 6.3.  Block = cvs x RB threads (cvs = 8-channel slots per row, RB rows side by side);
// grid = (row groups, images).
__global__ __launch_bounds__(320) void hv_affine_apply_kernel(const bf16_t* X, long ldx, int rows_per_image, int C, const bf16_t* X2,
                                                              long ldx2, int C2, const float* scale, const float* shift, int act,
                                                              bf16_t* Y, long ldy, int cvs, int rb) {
    const int cv = threadIdx.x % cvs, r0 = threadIdx.x / cvs;
    if (r0 >= rb) return;
    const int img = blockIdx.y;
    const long so = (long)img * (C + C2) + cv * 8;
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + so), s1 = *reinterpret_cast<const f32x4*>(scale + so + 4);
    const f32x4 t0 = *reinterpret_cast<const f32x4*>(shift + so), t1 = *reinterpret_cast<const f32x4*>(shift + so + 4);
    const bool second = cv * 8 >= C;
    const bf16_t* src = second ? X2 + (cv * 8 - C) : X + cv * 8;
    const long lds = second ? ldx2 : ldx;
    const long row0 = (long)img * rows_per_image;
    const int step = gridDim.x * rb;
    auto apply = [&](u32x4 v) __attribute__((always_inline)) {
        float f[8];
        hv_unpack8(v, f);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f[e] = hv_act(f[e] * s0[e] + t0[e], act);
            f[4 + e] = hv_act(f[4 + e] * s1[e] + t1[e], act);
        }
        return hv_pack8(f);
    };
    int r = blockIdx.x * rb + r0;
    for (; r + 3 * step < rows_per_image; r += 4 * step) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = hv_ld16(src + (row0 + r + u * step) * lds);
#pragma unroll
        for (int u = 0; u < 4; ++u) hv_st16(Y + (row0 + r + u * step) * ldy + cv * 8, apply(v[u]));
    }
    for (; r < rows_per_image; r += step) hv_st16(Y + (row0 + r) * ldy + cv * 8, apply(hv_ld16(src + (row0 + r) * lds)));
}

static inline void hv_affine_apply_launch(const bf16_t* X, long ldx, int rows, int rows_per_image, int C, const bf16_t* X2, long ldx2,
                                          int C2, const float* scale, const float* shift, int act, bf16_t* Y, long ldy,
                                          hipStream_t stream) {
    const int cvs = (C + C2) / 8;                       // <= 320 (checked by the callers: C + C2 <= 2560)
    const int rb = cvs >= 160 ? 1 : (cvs > 64 ? 2 : (256 / cvs));
    const int n_images = rows / rows_per_image;
    long gx = ((long)rows_per_image + 4 * rb - 1) / (4 * rb);  // ~ four rows per thread
    const long cap = (256L * 12 + n_images - 1) / n_images;     // ~ 12 blocks per CU over the whole grid
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    hv_note("hv_affine_apply_kernel | rows=%d C=%d C2=%d act=%d", rows, C, C2, act);
    hv_launch(hv_affine_apply_kernel, dim3((unsigned)gx, (unsigned)n_images), dim3((unsigned)((cvs * rb + 63) / 64 * 64)), stream, X, ldx,
              rows_per_image, C, X2, ldx2, C2, scale, shift, act, Y, ldy, cvs, rb);
}
