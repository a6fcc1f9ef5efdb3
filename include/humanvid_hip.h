/* humanvid_hip.h -- C ABI of libhumanvid_hip.so: the MI355X (gfx950) kernels of the CamAnimate
 * denoising path.
 *
 * The reference (zhenzhiwang/HumanVid) is pure Python and has no FFI / operator registry; its
 * boundary for this path is the Python class surface, which the `src.*` /
 * `humanvid_amd.*` host code reproduces.  This header is the native layer underneath it: every
 * entry point replaces a group of ATen / diffusers ops the reference dispatches, cited below as
 * /root/reference file:line.  Conventions:
 *   - plain C: raw device pointers, sizes, strides; the caller owns every buffer (including
 *     workspaces); the library keeps no state besides captured graphs.
 *   - activations: bfloat16 bits (uint16_t), channels-last  [image][y][x][channel]
 *     == [image][token][channel]; image index = batch * frames + frame.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on it.
 *   - return value: 0 on success, negative HV_E* code otherwise; hv_last_error() returns a
 *     thread-local description of the last failure.
 */
#ifndef HUMANVID_HIP_H
#define HUMANVID_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HV_OK 0
#define HV_EINVAL (-1)  /* shape / alignment / dtype violation */
#define HV_ENOTSUP (-2) /* e.g. head dim not in {40, 80, 160} */
#define HV_EHIP (-3)    /* HIP runtime error, see hv_last_error() */

#define HV_ACT_NONE 0
#define HV_ACT_SILU 1
#define HV_ACT_RELU 2

const char* hv_last_error(void);
int hv_abi_version(void);
/* sizeof() of every parameter struct, in declaration order -- lets a binding verify its mirror */
int hv_struct_sizes(int* out, int capacity);

/* ---- dense contraction --------------------------------------------------------------------
 * Y[M,N] = epi( pro(X)[M,K] . W[N,K]^T ).  Replaces nn.Linear / 1x1 nn.Conv2d call sites:
 * Transformer3DModel proj_in/out (src/models/transformer_3d.py:125-166), motion-module
 * proj_in/out (src/models/motion_module.py:157-175), diffusers Attention to_q/k/v/to_out,
 * FeedForward/GEGLU, TimestepEmbedding (src/models/unet_3d.py:461-467), time_emb_proj
 * (src/models/resnet.py:224), conv_shortcut (src/models/resnet.py:211-213).               */
typedef struct hv_gemm_params {
    const uint16_t* X;
    long ldx;
    const uint16_t* X2; /* optional: columns k >= K1 come from X2[m][k-K1] (skip concat) */
    long ldx2;
    int K1;
    const uint16_t* W; /* [N][K] */
    void* Y;           /* bf16, or float when out_f32 */
    long ldy;
    int out_f32;
    uint16_t* Yt; /* columns n >= n_split are stored transposed: Yt[(n-n_split)*ldyt + m] */
    long ldyt;
    int n_split;
    int M, N, K;
    const float* pro_scale; /* [M/rows_per_image][K]: x' = act(x*scale + shift) (GroupNorm apply) */
    const float* pro_shift;
    int rows_per_image;
    int pro_act;
    const float* bias;     /* [N] */
    const float* row_mean; /* LayerNorm fold: y = rstd[m]*(acc - mean[m]*colsum[n]) + bias[n] */
    const float* row_rstd;
    const float* colsum;
    const float* pe; /* [pe_frames][N], row m uses frame (m / pe_period) % pe_frames */
    int pe_period;
    int pe_frames;
    const float* rowvec; /* [M/rowvec_period][N] (time embedding, folded cross-attention) */
    int rowvec_period;
    const uint16_t* residual;
    long ldr;
    int geglu; /* W rows packed as [16 h | 16 g] blocks; Y has N/2 columns: h * gelu(g) */
    int out_act;
    /* optional row permutation of the OUTPUT (and of the residual read): with perm_p > 0 the row index
     * m = (x * perm_y + y) * perm_p + p is stored at (y * perm_x + x) * perm_p + p -- the transpose of the two outer axes
     * of an [x][y][perm_p] row index.  Lets the frame-sharded motion module write the all-to-all send layout straight
     * from the QKV projection and fold the inverse re-ordering into the output projection. */
    int perm_x, perm_y, perm_p;
    /* optional GroupNorm partial statistics of the OUTPUT (round 3): per (image, part, channel) {sum, sum of squares} of the
     * stored values, fp32 [M / gn_rows_per_image][parts][N][2], parts = hv_gemm_gn_parts(p) row ranges per image -- what
     * hv_groupnorm_from_parts turns into the scale / shift of the next GroupNorm, so that the activation is not read again
     * by a statistics pass.  NULL: not wanted.  Only for problems hv_gemm_gn_parts() accepts (plain bf16 output on the
     * 128x128x64 LDS-DMA kernel, whole wave sub-tiles per image). */
    float* gn_part;
    int gn_rows_per_image;
    /* optional LayerNorm partial statistics of the OUTPUT rows: fp32 [M][hv_gemm_ln_parts(p)][2] = {sum, sum of squares} of
     * the stored values over each column block of a row (blocks of 64 columns from the tile kernels, of 80 from hv_gemm_wr_kernel:
     * the count is what hv_gemm_ln_parts reports, the consumer only adds them up); hv_layernorm_from_parts turns them into the row mean / rstd that
     * the next LayerNorm-folded GEMM reads -- no statistics pass over the activation.  NULL: not wanted; not together with
     * gn_part; same kernel restriction as gn_part. */
    float* ln_part;
} hv_gemm_params;
int hv_gemm_gn_parts(const hv_gemm_params* p); /* parts per image hv_gemm would write for this problem, 0 = cannot */
int hv_gemm_ln_parts(const hv_gemm_params* p); /* column blocks per row hv_gemm would write (N / 64 from the tile kernels, 4 from hv_gemm_wr_kernel), 0 = cannot */
int hv_gemm(const hv_gemm_params* p, void* stream);

/* ---- 3x3 convolution (implicit GEMM, LDS halo tile) ---------------------------------------
 * Replaces InflatedConv3d 3x3 (src/models/resnet.py:9-15) incl. ResnetBlock3D conv1/conv2 with the
 * preceding InflatedGroupNorm+SiLU applied on load (src/models/resnet.py:215-245), Downsample3D
 * (stride 2, :110-118), Upsample3D (nearest 2x folded into addressing, :51-88), the skip
 * torch.cat (src/models/unet_3d_blocks.py:698,828: two-source channel loop), PoseGuider convs
 * (src/models/pose_guider.py:51-61) and the camera encoder convs (src/cameractrl/pose_adaptor.py). */
#define HV_CONV_S1 0
#define HV_CONV_S2 1
#define HV_CONV_UP2 2
typedef struct hv_conv3x3_params {
    const uint16_t* X; /* [n][Hs][Ws][C1] */
    int C1;
    const uint16_t* X2; /* optional second source, channels C1..C1+C2 */
    int C2;
    const uint16_t* W; /* packed [Cout][9][C1+C2], tap = ky*3+kx */
    uint16_t* Y;       /* [n][Ho][Wo][Cout] */
    int n_images, Hs, Ws, Ho, Wo, Cout;
    int mode;
    const float* pro_scale; /* [n][Cin] GroupNorm apply (+ pro_act) on load; padding stays zero */
    const float* pro_shift;
    int pro_act;
    const float* bias;
    const float* rowvec; /* time embedding: row (n/images_per_rowvec), stride rowvec_ld floats */
    int images_per_rowvec;
    long rowvec_ld;
    const uint16_t* residual; /* [residual_images][Ho][Wo][Cout], image index taken modulo */
    int residual_images;
    int out_act;
    /* optional GroupNorm partial statistics of the OUTPUT: fp32 [n_images][parts][Cout][2] = {sum, sum of squares} over the
     * pixels of one wave's share of an output patch, parts = hv_conv3x3_gn_parts(p); NULL: not wanted (see hv_gemm_params) */
    float* gn_part;
} hv_conv3x3_params;
int hv_conv3x3(const hv_conv3x3_params* p, void* stream);
int hv_conv3x3_gn_parts(const hv_conv3x3_params* p); /* parts per image for this problem (kernel selection included) */

/* ---- GroupNorm statistics -> per-(image,channel) affine -----------------------------------
 * InflatedGroupNorm (src/models/resnet.py:18-26), torch.nn.GroupNorm in Transformer3DModel
 * (src/models/transformer_3d.py:58-60,123) and the motion module (src/models/motion_module.py:119).
 * Produces scale/shift so that GN(x)[c] = x*scale[img][c] + shift[img][c]; the apply step is fused
 * into the consumer (hv_conv3x3 / hv_gemm prologue).                                       */
typedef struct hv_groupnorm_params {
    const uint16_t* X;
    int C1;
    const uint16_t* X2;
    int C2;
    int n_images, pixels, groups;
    float eps;
    const float* gamma;
    const float* beta;
    float* partial; /* workspace: n_images * splits * groups * 2 floats */
    int splits;     /* pixel ranges per image, 1 .. 64 (one lane of the merging wavefront each) */
    float* scale; /* out [n_images][C] */
    float* shift;
} hv_groupnorm_params;
int hv_groupnorm_affine(const hv_groupnorm_params* p, void* stream);

/* The same scale / shift from the partial statistics the PRODUCING kernels left (hv_conv3x3 / hv_gemm gn_part): no pass over
 * the activation.  Two sources = the channel concat of the up-blocks (groups may straddle the seam).  Sums are merged in
 * double precision: mean = S / n, var = Q / n - mean^2 over the group's channels and all parts. */
typedef struct hv_gn_parts_params {
    const float* part1; /* [n_images][parts1][C1][2] */
    int parts1, C1;
    const float* part2; /* optional second source [n_images][parts2][C2][2] */
    int parts2, C2;
    int n_images, pixels, groups;
    float eps;
    const float* gamma;
    const float* beta;
    float* scale; /* out [n_images][C1 + C2] */
    float* shift;
} hv_gn_parts_params;
int hv_groupnorm_from_parts(const hv_gn_parts_params* p, void* stream);

/* LayerNorm row statistics (nn.LayerNorm eps 1e-5; src/models/attention.py:329-360,
 * src/models/motion_module.py:228,234); normalisation itself is folded into hv_gemm.       */
int hv_layernorm_stats(const uint16_t* X, long ldx, int M, int C, float eps, float* mean, float* rstd,
                       void* stream);
/* the same mean / rstd from the partial row sums the producing hv_gemm left (ln_part: [M][parts][2]; any number of parts) */
int hv_layernorm_from_parts(const float* part, int parts, int M, int C, float eps, float* mean, float* rstd, void* stream);

/* ---- spatial self-attention with reference-bank keys --------------------------------------
 * diffusers Attention/AttnProcessor2_0 SDPA as used by the patched TemporalBasicTransformerBlock
 * (src/models/mutual_self_attention.py:147-186): per image, keys/values = own tokens, followed by
 * the reference bank tokens for images whose bank_sel >= 0 (the CFG-unconditional half attends
 * to its own tokens only).  Flash-style online softmax, MFMA 16x16x32.                      */
typedef struct hv_attention_params {
    const uint16_t* Q; /* row = img*Lq + q, head h at column h*D */
    long ldq;
    const uint16_t* K; /* row = img*L1 + kv */
    long ldk;
    const uint16_t* Vt; /* transposed values: Vt[(h*D+d)*ldvt + img*L1 + kv] */
    long ldvt;
    const uint16_t* K2; /* bank keys: row = sel*L2 + kv */
    long ldk2;
    const uint16_t* Vt2;
    long ldvt2;
    const int* bank_sel; /* [n_images] or NULL */
    uint16_t* O;
    long ldo;
    int n_images, heads, D, Lq, L1, L2;
    float scale;
} hv_attention_params;
int hv_attention(const hv_attention_params* p, void* stream);

/* fp8 (OCP e4m3) form of hv_attention -- BASELINE.json configs[4]: QK^T and PV on v_mfma_f32_16x16x32_fp8_fp8 with fp32
 * accumulation; same semantics as hv_attention (transposed-V form).  Keys and values are quantised once per call by
 *   hv_attention_fp8_quantize (one key source per call: the n_images images with L = L1, or the bank batches with L = L2):
 *     phase & 1: kscale[(img*heads + h)*ceil(L/64) + tile] = amax(K tile) / 384,  vamax[h] = amax over all images of V;
 *     phase & 2: K8[(img*L + kv)*ldk8 + h*D + d] = e4m3(K / kscale),  Vt8[(h*D + d)*ldvt8 + img*L + kv] = e4m3(V / vscale),
 *                vscale = max(vamax[h], vfloor ? vfloor[h] : 0) / 384 -- pass the OTHER key source's vamax as vfloor so that own
 *                values and bank values carry one scale per head (amax both sources first, then quantise both);
 *   hv_attention_fp8: p->K / Vt / K2 / Vt2 point at the e4m3 tensors (ld* in bytes = elements), p->Q is bf16 (scaled per
 *     query row inside), kscale / vamax of the own keys, kscale2 / vamax2 of the bank.
 * Stated accuracy: NRMSE <= 8e-2 against fp32 softmax attention on i.i.d. random bf16 operands (four e4m3 roundings of
 * 2^-4 / sqrt 3 rms each; bf16 kernel: <= 6e-3); denoiser output with fp8 attention <= 3e-2 against the fp32 oracle. */
int hv_attention_fp8_quantize(const uint16_t* K, long ldk, const uint16_t* Vt, long ldvt, int n_images, int heads, int D, int L,
                              float* kscale, float* vamax, const float* vfloor, uint8_t* K8, long ldk8, uint8_t* Vt8, long ldvt8,
                              int phase, void* stream);
int hv_attention_fp8(const hv_attention_params* p, const float* kscale, const float* vamax, const float* kscale2,
                     const float* vamax2, void* stream);

/* kernel-variant selection for A/B measurements (process-global; not needed for correctness) */
#define HV_TUNE_ATTN_D40 0     /* head dim 40: 0 (default) = the dedicated 8-wave kernel (hv_attention40.h: 32x32x16 QK^T, 256 queries per workgroup); 2 = the generic kernel (A/B) */
#define HV_TUNE_GEMM_MAX_GRID 2 /* persistent GEMM workgroups (multiple of 8, default 512) */
#define HV_TUNE_GEMM_GLDS 3     /* GEMM kernel selection: 1 = default -- LDS-DMA kernels: 256x320x64 wide tiles for N = 320, K >= 640, M % 256 == 0 (plain-output forms); otherwise 256x256x64 tiles (one 8-wave workgroup per CU) when N >= 960 and the tiles fill the last round over the 256 CUs to >= 90 %, 128x128x64 (two 4-wave workgroups per CU) otherwise; 2 = 256x256x64 wherever the shape allows, 3 = 128x128x64 everywhere, 0 = register-staged kernel (A/Bs; results are bit-identical across 0 / 2 / 3 / 6); 6 = no wide tiles (the round-3 default) */
#define HV_TUNE_GEMM_W4 10      /* which problems of the 256x256x64 class run on four waves (hv_gemm4.h: one wave per SIMD, 192x256x64 tiles, deferred output stores): 1 (default) = the LayerNorm-fold forms (with / without GEGLU) at K >= 640, M % 192 == 0; 0 = none; 2 = every problem whose shape allows it (M % 256 == 0, N % 64 == 0, one X source), stores at once; 3 = 2 + deferred stores where the form has them (K >= 320); 4 = 3 on the plain tile raster (A/Bs and tests; results are bit-identical) */
#define HV_TUNE_GEMM_XS 11      /* 1: the LayerNorm-fold forms (with / without GEGLU) at K = 320, M % 192 == 0 run on the X-stationary kernel (hv_gemm_xs.h: the row block's X resident in LDS, only W streams, the epilogue software-pipelined into the next tile); 0: the 8-wave 256x256x64 kernel (results are bit-identical) */
#define HV_TUNE_GEMM_C4 13      /* bias (+ residual) projections with N % 320 == 0, M % 192 == 0 on hv_gemm_c4_kernel (hv_gemm_c4.h: 192x320x64 tiles on four waves, the k-loop of hv_conv_w4_kernel): 1 (default) = K >= 1280, tiles that fill their rounds of 256 CUs to >= 70 % (>= 128 tiles), N % 256 != 0 (feed-forward output projections of levels 0 and 1); 0 = never; 2 = wherever the structure allows (tests; results are bit-identical) */
#define HV_TUNE_GEMM_WR 15      /* K = 320: N = 320 bias (+ table row) (+ residual) projections, with or without statistics of the output, and N = 320 / 640 / 960 LayerNorm-fold projections (+ table row) on hv_gemm_wr_kernel (hv_gemm_wr.h: weights in registers, X and residual streamed through LDS rings): 1 (default) = M >= 16 384 (M % 64 == 0), 0 = never, 2 = always (tests; Y is bit-identical, hv_gemm_ln_parts() reports 4 parts of 80 columns for it) */
#define HV_TUNE_CONV_GLDS 4     /* 1: conv weight tiles by LDS-DMA (default), 0: register-staged */
#define HV_TUNE_CONV_BIG 5      /* 1 (default): 256-pixel tiles for the upsample-folded convolution, 64-channel reduction chunks for stride-1 convolutions on images of <= 384 pixels; 0 / 2: neither / only the 256-pixel tiles (A/Bs); 3: 64-channel chunks for every stride-1 convolution whose sources allow them (A/B) */
#define HV_TUNE_CONV_RASTER 9    /* workgroup raster of hv_conv3x3 inside an XCD: 0 = the output-channel tiles of a pixel patch adjacent (halo shared through L2, weights re-streamed), 1 = the pixel patches of an output-channel tile adjacent (weights stay in L2), 2 (default) = 1 where Cin x Cout >= 640 x 640: HBM-side fetch of the 1280 -> 1280 convolution 1.4 -> 0.6 GB per launch at equal time (profiles/r04_s3.txt) */
#define HV_TUNE_CONV_W4 12       /* stride-1 convolutions of a plain single-source input with Cout % 320 == 0 or Cout % 256 == 0 on hv_conv_w4_kernel (hv_conv4.h: 12 x 16 pixels x 320 / 256 channels on four waves, halo and weights by LDS-DMA; the width whose tiles fill the rounds of 256 CUs better): 1 (default) = where the tiles fill their rounds of 256 CUs to >= 70 % (>= 128 tiles) and the patches cover >= 80 % of the image, 0 = never, 2 = wherever the structure allows (tests), 3 = as 1 with 320-channel tiles only (A/B) */
#define HV_TUNE_CMDLIST_GRAPHS 7 /* 1 (default): hv_cmdlist_run replays a recorded segment as ONE captured HIP graph; 0: re-issues its launches one by one (A/B) */
int hv_set_tuning(int key, int value);

/* ---- temporal self-attention over the frame axis ------------------------------------------
 * VersatileAttention (src/models/motion_module.py:351-388) and the camera encoder's
 * TemporalSelfAttention (src/cameractrl/motion_module.py:323-388): for every (batch, pixel, head)
 * attention of the Fq local query frames over all Fkv frames (Fq < Fkv when the clip is sharded
 * along the frame axis across GPUs and K/V were all-gathered).                                   */
typedef struct hv_temporal_attention_params {
    const uint16_t* Q; /* row (b*Fq + fq)*P + p, head h at column h*D */
    long ldq;
    const uint16_t* K; /* row b*kv_stride_b + (f / kv_chunk)*kv_stride_chunk + (f % kv_chunk)*P + p */
    const uint16_t* V;
    long ldkv;
    long kv_stride_b;     /* in rows */
    long kv_stride_chunk; /* in rows; frame-sharded runs gather K/V as [rank][b][F/ranks][P] */
    int kv_chunk;
    uint16_t* O; /* same row order as Q */
    long ldo;
    int B, Fq, Fkv, P, heads, D;
    float scale;
    int qo_chunked; /* 1: Q and O rows follow the K / V row formula (needs Fq == Fkv): every operand lives in the
                       [rank][b][F/ranks][P] layout an all-to-all leaves behind -- no re-ordering copies around the kernel */
} hv_temporal_attention_params;
int hv_temporal_attention(const hv_temporal_attention_params* p, void* stream);

/* ---- layout adaptors at the drop-in boundary ----------------------------------------------
 * [b][c][f][h][w] (fp32 or bf16, the reference's layout) <-> [(rep b) f][h][w][cpad] bf16.     */
/* frames: optional DEVICE int[F] of source frame indices (context window / frame shard); NULL = 0..F-1 */
int hv_pack_ncfhw(const void* src, int src_is_bf16, int B, int C, int Fsrc, int H, int W, const int* frames, int F,
                  int rep, uint16_t* dst, int Cpad, void* stream);
int hv_unpack_nhwc(const uint16_t* src, int ldc, int B, int C, int F, int H, int W, void* dst, int dst_is_bf16,
                   void* stream);
/* nn.PixelUnshuffle(r) on [b][c][f][H][W] fp32 -> [(b f)][H/r][W/r][c*r*r] bf16
 * (src/cameractrl/pose_adaptor.py:177,236) */
int hv_pixel_unshuffle(const float* src, int B, int C, int F, int H, int W, int r, uint16_t* dst, void* stream);
/* ray_condition (src/dataset/dance_image_h_v_camera.py:88-130) fused with that PixelUnshuffle: Pluecker map (o x d, d)
 * from per-frame intrinsics K [F][4] = (fx, fy, cx, cy) in pixels and camera-to-world matrices c2w [F][4][4] (fp32,
 * device memory), written as [F][H/r][W/r][6*r*r] bf16 -- the camera encoder's input -- without materialising the
 * [1,6,F,H,W] fp32 map */
int hv_plucker_unshuffle(const float* K, const float* c2w, int F, int H, int W, int r, uint16_t* dst, void* stream);
/* diffusers Timesteps(320, flip_sin_to_cos=True, shift 0) (src/models/unet_3d.py:93,461): [B][dim] bf16 */
/* GroupNorm apply as its own pass: Y[row][c] = act(X[row][c] * scale[row / rows_per_image][c] + shift[...][c]), bf16 -> bf16.
 * Replaces the normalisation half of the InflatedGroupNorm in front of Transformer3DModel.proj_in
 * (src/models/transformer_3d.py:125-131) and TemporalTransformer3DModel.proj_in (src/models/motion_module.py:157-163);
 * scale / shift come from hv_groupnorm_affine. */
int hv_affine_apply(const uint16_t* X, long ldx, int rows, int rows_per_image, int C, const float* scale, const float* shift,
                    int act, uint16_t* Y, long ldy, void* stream);
/* The same pass over the channel concatenation [X | X2] of a decoder ResnetBlock3D's input (torch.cat([hidden_states,
 * res_hidden_states], dim=1) in src/models/unet_3d_blocks.py CrossAttnUpBlock3D / UpBlock3D.forward, normalised by
 * ResnetBlock3D.norm1 + SiLU, src/models/resnet.py:215-222): Y [rows][C + C2], scale / shift [images][C + C2].  The 3x3
 * convolution behind it then reads ONE normalised source (hv_conv3x3 without pro_scale): the convolution's own prologue
 * repeats the transform in every 128-channel output tile and in every halo pixel. */
int hv_affine_apply_cat(const uint16_t* X, long ldx, int C, const uint16_t* X2, long ldx2, int C2, int rows, int rows_per_image,
                        const float* scale, const float* shift, int act, uint16_t* Y, long ldy, void* stream);
int hv_timestep_embedding(const float* t, int B, int dim, uint16_t* dst, void* stream);

/* ---- window accumulation, classifier-free guidance and the DDIM v-prediction update --------
 * src/pipelines/pipeline_pose2vid_long.py:550-563 (+ diffusers DDIMScheduler.step, eta = 0).
 * pred: [(rep f_win)][h][w][ldc] bf16 (conv_out output); acc: fp32 [rep][C][F][h][w]; counter [F]. */
int hv_accumulate_window(const uint16_t* pred, int ldc, int rep, int C, int f_win, int H, int W, const int* frames,
                         int F, float* acc, float* counter, void* stream);
/* coeffs: DEVICE pointer to {guidance, sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)} so that
 * a captured step graph can be replayed for every timestep */
int hv_cfg_ddim_step(float* latents, float* acc, float* counter, int rep, int C, int F, int H, int W,
                     const float* coeffs, void* stream);

/* ---- HIP graph capture of a launch sequence (one denoising step) ---------------------------- */
int hv_graph_begin(void* stream);
int hv_graph_end(void* stream, void** graph_exec_out);
int hv_graph_launch(void* graph_exec, void* stream);
int hv_graph_destroy(void* graph_exec);

/* ---- command lists: record a launch sequence once, re-issue it with one native loop ------------
 * While recording, launches execute normally AND are appended to the list.  hv_cmdlist_cut() closes
 * the current segment and opens the next (used at every RCCL collective of a frame-sharded step). */
int hv_cmdlist_begin(void);
int hv_cmdlist_cut(void** list_out);
int hv_cmdlist_end(void** list_out);
int hv_cmdlist_size(void* list);
int hv_cmdlist_run(void* list, void* stream);
int hv_cmdlist_fallbacks(void); /* command lists whose graph capture failed so far (they are re-issued launch by launch) */
int hv_cmdlist_destroy(void* list);

/* ---- launch profile -----------------------------------------------------------------------------
 * Between hv_profile_begin() and hv_profile_end() every kernel launch of this thread is bracketed by two HIP events
 * on its own stream.  hv_profile_end() waits for them and writes one text line per distinct "kernel variant | shape"
 * key: "<launches>\t<total milliseconds>\t<key>\n", followed by one line "#seq\t<i>,<i>,..." = the launch order as
 * 0-based indices into those lines.  Returns the buffer size needed, negative on error.  When the
 * size exceeds `capacity` (or out == NULL, capacity >= 0: a size query) nothing is written, recording has stopped and the
 * text is kept: call again with a buffer of the returned size to collect it (out == NULL with capacity < 0 drops it).  bench.py builds its roofline block from one profiled,
 * eagerly launched denoising step (the real epilogues / multiplicities / variants of the step). */
int hv_profile_begin(void);
int hv_profile_end(char* out, int capacity);

/* timing helper used by bench.py: elapsed milliseconds between two events it records on `stream` */
int hv_event_create(void** ev);
int hv_event_record(void* ev, void* stream);
int hv_event_elapsed_ms(void* start, void* stop, float* ms);
int hv_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif
