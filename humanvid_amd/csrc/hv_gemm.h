// hv_gemm.h -- bf16 MFMA GEMM  Y[M,N] = epilogue( prologue(X)[M,K] . W[N,K]^T )  for gfx950.
//
// Serves every dense contraction of the denoising path that is not a 3x3 convolution:
// 1x1 convs / Linear proj_in/out, fused QKV, attention out-projection, GEGLU feed-forward,
// time-embedding MLP.  Reference call sites: src/models/transformer_3d.py:125-166,
// src/models/motion_module.py:157-175, diffusers Attention/FeedForward.
//
// Design (MI355X):
//  * 128x128 output tile per 256-thread workgroup (4 waves, 2x2, 64x64 per wave = 4x4 MFMA
//    16x16x32 fragments, 64 fp32 accumulator VGPRs), BK = 64, LDS double-buffered (64 KiB) with one
//    barrier per K-step; the next tile's global loads are issued before the MFMA block so HBM
//    latency hides under 32 MFMAs per wave.
//  * both operands are K-contiguous ("B^T" form), so a fragment is one 16-byte LDS read per
//    lane; rows are 128 B, XOR-swizzled over (row>>1)&7 so the 16 lanes of a ds_read_b128 group
//    hit 16 distinct 16-byte slots.
//  * operands are swapped (A = weight rows, B = activation rows): a lane then owns 4 consecutive
//    output channels of one token -> 8-byte stores into the row-major output and vector
//    loads of the per-channel epilogue terms.
//  * fused prologue on X: per-(image,channel) affine (GroupNorm apply) + optional SiLU.
//  * fused epilogue: LayerNorm folded algebraically (row mean/rstd + column sums), bias,
//    positional-encoding table, per-batch row vector (time embedding / folded cross-attention),
//    GEGLU gate, residual add, activation, and an optional transposed store of a column range
//    (writes V^T for the attention kernel).
//  * workgroup ids are remapped so that the tiles sharing an X row-panel run on one XCD (L2).
#pragma once
#include "hv_common.h"
#include "humanvid_hip.h"

typedef hv_gemm_params HvGemmParams;  // declared in include/humanvid_hip.h

template <int BK>
HV_DEV int hv_swz(int row, int chunk) {
    constexpr int CPR = BK / 8;     // 16-byte chunks per row
    constexpr int RPB = 16 / CPR;   // rows per 256-byte bank row
    return row * (BK * 2) + ((chunk ^ ((row / RPB) % CPR)) << 4);
}

template <int BK>
__global__ __launch_bounds__(256) void hv_gemm_kernel(HvGemmParams p) {
    constexpr int BM = 128, BN = 128;
    constexpr int CPR = BK / 8;
    constexpr int CH_PER_THREAD = (BM * CPR) / 256;  // chunks of one operand tile per thread
    constexpr int TILE_BYTES = BM * BK * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * TILE_BYTES];
    unsigned char* Xs = smem;                    // [2][BM][BK]
    unsigned char* Ws = smem + 2 * TILE_BYTES;   // [2][BN][BK]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int r16 = lane & 15, quad = lane >> 4;

    // XCD-aware tile mapping: all N-tiles of one M-panel stay on the XCD that dispatched them
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int total = tiles_n * tiles_m;
    const int cpx = gridDim.x / 8;
    const int t = (blockIdx.x % 8) * cpx + blockIdx.x / 8;
    if (t >= total) return;
    const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;

    u32x4 xr[CH_PER_THREAD], wr[CH_PER_THREAD];
    const int nk = p.K / BK;

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < CH_PER_THREAD; ++i) {
            const int id = tid + 256 * i;
            const int row = id / CPR, c = id % CPR;
            const int m = m0 + row, n = n0 + row, k = k0 + c * 8;
            u32x4 z = {0u, 0u, 0u, 0u};
            xr[i] = z;
            wr[i] = z;
            if (m < p.M) {
                const bf16_t* src = (p.X2 != nullptr && k >= p.K1) ? p.X2 + (long)m * p.ldx2 + (k - p.K1)
                                                                   : p.X + (long)m * p.ldx + k;
                xr[i] = hv_ld16(src);
            }
            if (n < p.N) wr[i] = hv_ld16(p.W + (long)n * p.K + k);
        }
    };

    auto store_tile = [&](int kt, int buf) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < CH_PER_THREAD; ++i) {
            const int id = tid + 256 * i;
            const int row = id / CPR, c = id % CPR;
            u32x4 xv = xr[i];
            if (p.pro_scale != nullptr || p.pro_act != HV_ACT_NONE) {
                const int m = m0 + row;
                if (m < p.M) {
                    float f[8];
                    hv_unpack8(xv, f);
                    if (p.pro_scale != nullptr) {
                        const long o = (long)(m / p.rows_per_image) * p.K + k0 + c * 8;
                        const f32x4 s0 = *reinterpret_cast<const f32x4*>(p.pro_scale + o);
                        const f32x4 s1 = *reinterpret_cast<const f32x4*>(p.pro_scale + o + 4);
                        const f32x4 t0 = *reinterpret_cast<const f32x4*>(p.pro_shift + o);
                        const f32x4 t1 = *reinterpret_cast<const f32x4*>(p.pro_shift + o + 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            f[j] = f[j] * s0[j] + t0[j];
                            f[4 + j] = f[4 + j] * s1[j] + t1[j];
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = hv_act(f[j], p.pro_act);
                    xv = hv_pack8(f);
                }
            }
            hv_st16(Xs + buf * TILE_BYTES + hv_swz<BK>(row, c), xv);
            hv_st16(Ws + buf * TILE_BYTES + hv_swz<BK>(row, c), wr[i]);
        }
    };

    f32x4 acc[4][4];  // [nf][mf]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_tile(0);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        store_tile(kt, buf);
        __syncthreads();
        if (kt + 1 < nk) load_tile(kt + 1);
        const unsigned char* xs = Xs + buf * TILE_BYTES;
        const unsigned char* ws = Ws + buf * TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8 wf[4], xf[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                wf[f] = hv_as_bf16x8(hv_ld16(ws + hv_swz<BK>(64 * wn + 16 * f + r16, kk * 4 + quad)));
                xf[f] = hv_as_bf16x8(hv_ld16(xs + hv_swz<BK>(64 * wm + 16 * f + r16, kk * 4 + quad)));
            }
#pragma unroll
            for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                for (int mf = 0; mf < 4; ++mf)
                    acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nf], xf[mf], acc[nf][mf], 0, 0, 0);
        }
    }

    // ---- epilogue: lane owns token m (column of the MFMA tile) and 4 consecutive channels n
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) {
        const int m = m0 + 64 * wm + 16 * mf + r16;
        if (m >= p.M) continue;
        float mean = 0.f, rstd = 1.f;
        if (p.row_rstd != nullptr) {
            mean = p.row_mean[m];
            rstd = p.row_rstd[m];
        }
        const float* pe_row = p.pe ? p.pe + (long)((m / p.pe_period) % p.pe_frames) * p.N : nullptr;
        const float* rv_row = p.rowvec ? p.rowvec + (long)(m / p.rowvec_period) * p.N : nullptr;
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const int n = n0 + 64 * wn + 16 * nf + 4 * quad;
            if (n >= p.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = acc[nf][mf][r];
                if (p.row_rstd != nullptr) a = rstd * (a - mean * p.colsum[n + r]);
                if (p.bias != nullptr) a += p.bias[n + r];
                if (pe_row != nullptr) a += pe_row[n + r];
                if (rv_row != nullptr) a += rv_row[n + r];
                v[r] = a;
            }
            if (p.geglu) {
                // packed weight rows: [16 x h | 16 x g] blocks -> fragment pairs (even nf: h, odd nf: g)
                acc[nf][mf] = f32x4{v[0], v[1], v[2], v[3]};
                if ((nf & 1) == 0) continue;
                const int no = ((n0 + 64 * wn + 16 * (nf - 1)) >> 1) + 4 * quad;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[nf - 1][mf][r] * hv_gelu_erf(v[r]);
                if (p.residual != nullptr) {
                    const u32x2 rr = hv_ld8(p.residual + (long)m * p.ldr + no);
                    v[0] += hv_bf2f((bf16_t)(rr[0] & 0xffff));
                    v[1] += hv_bf2f((bf16_t)(rr[0] >> 16));
                    v[2] += hv_bf2f((bf16_t)(rr[1] & 0xffff));
                    v[3] += hv_bf2f((bf16_t)(rr[1] >> 16));
                }
                u32x2 o = {hv_pack2(v[0], v[1]), hv_pack2(v[2], v[3])};
                hv_st8(reinterpret_cast<bf16_t*>(p.Y) + (long)m * p.ldy + no, o);
                continue;
            }
            if (p.residual != nullptr) {
                const u32x2 rr = hv_ld8(p.residual + (long)m * p.ldr + n);
                v[0] += hv_bf2f((bf16_t)(rr[0] & 0xffff));
                v[1] += hv_bf2f((bf16_t)(rr[0] >> 16));
                v[2] += hv_bf2f((bf16_t)(rr[1] & 0xffff));
                v[3] += hv_bf2f((bf16_t)(rr[1] >> 16));
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = hv_act(v[r], p.out_act);
            if (p.Yt != nullptr && n >= p.n_split) {
#pragma unroll
                for (int r = 0; r < 4; ++r) p.Yt[(long)(n - p.n_split + r) * p.ldyt + m] = hv_f2bf(v[r]);
            } else if (p.out_f32) {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.Y) + (long)m * p.ldy + n) =
                    f32x4{v[0], v[1], v[2], v[3]};
            } else {
                u32x2 o = {hv_pack2(v[0], v[1]), hv_pack2(v[2], v[3])};
                hv_st8(reinterpret_cast<bf16_t*>(p.Y) + (long)m * p.ldy + n, o);
            }
        }
    }
}

static inline int hv_gemm_launch(const HvGemmParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return -1;
    if (p.K % 64 != 0 || p.N % 4 != 0) return -1;
    if (p.X2 != nullptr && (p.K1 % 64 != 0)) return -1;
    if (p.geglu && (p.N % 32 != 0 || p.Yt != nullptr || p.out_f32)) return -1;
    if (p.Yt != nullptr && (p.n_split % 16 != 0)) return -1;
    const int tiles = ((p.N + 127) / 128) * ((p.M + 127) / 128);
    const int grid = ((tiles + 7) / 8) * 8;
    hv_launch(hv_gemm_kernel<64>, dim3(grid), dim3(256), stream, p);
    return 0;
}
