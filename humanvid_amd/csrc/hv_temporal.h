// hv_temporal.h -- temporal self-attention over the frame axis (AnimateDiff motion module).
//
// Reference: VersatileAttention.forward (/root/reference/src/models/motion_module.py:351-388):
// '(b f) d c -> (b d) f c', SDPA over the f axis with 8 heads, back; and the camera encoder's
// TemporalSelfAttention (/root/reference/src/cameractrl/motion_module.py:323-388).  The positional
// encoding add and the LayerNorm before it are folded into the QKV GEMM (hv_gemm epilogue), so the
// kernel sees projected q|k|v rows.
//
// 0.2 % of the step's FLOPs but a sequence length of only 24: the op is HBM-bound.
// The rearranges of the reference disappear into addressing: rows of one pixel are F strided rows
// of the token-major [(b f) p][3C] tensor.  One workgroup handles one (batch, pixel) for all
// heads: K and V (F x C each) are staged once in LDS with coalesced 16-byte loads (each
// (frame,pixel) row is a contiguous 3C-element run), then thread (head, query frame) computes its
// F scores, softmax and output in fp32 registers; K/V reads are LDS broadcasts across the
// query-frame lanes of a head.
#pragma once
#include "hv_common.h"
#include "humanvid_hip.h"

template <int D, int FMAX>
__global__ __launch_bounds__(256) void hv_temporal_kernel(hv_temporal_attention_params p) {
    constexpr int HEADS = 8;
    constexpr int C = HEADS * D;
    __shared__ __attribute__((aligned(16))) bf16_t Ksm[FMAX * C];
    __shared__ __attribute__((aligned(16))) bf16_t Vsm[FMAX * C];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int F = p.Fkv, FQ = p.Fq;
    const int b = blockIdx.x / p.P, pix = blockIdx.x % p.P;
    const long row0 = ((long)b * FQ) * p.P + pix;  // query / output row of local frame f: row0 + f * P

    const int cv = C / 8;
    for (int i = tid; i < F * cv; i += nthr) {
        const int f = i / cv, c = i % cv;
        const long kvrow = (long)b * p.kv_stride_b + (long)(f / p.kv_chunk) * p.kv_stride_chunk +
                           (long)(f % p.kv_chunk) * p.P + pix;
        hv_st16(Ksm + f * C + c * 8, hv_ld16(p.K + kvrow * p.ldkv + c * 8));
        hv_st16(Vsm + f * C + c * 8, hv_ld16(p.V + kvrow * p.ldkv + c * 8));
    }
    __syncthreads();

    const int h = tid / FQ, fq = tid % FQ;
    if (h >= HEADS) return;
    const bf16_t* qrow = p.Q + (row0 + (long)fq * p.P) * p.ldq + h * D;

    float s[FMAX];
#pragma unroll
    for (int kf = 0; kf < FMAX; ++kf) s[kf] = 0.f;
#pragma unroll
    for (int dc = 0; dc < D / 8; ++dc) {
        float q8[8];
        hv_unpack8(hv_ld16(qrow + dc * 8), q8);
#pragma unroll
        for (int kf = 0; kf < FMAX; ++kf) {
            if (kf < F) {
                float k8[8];
                hv_unpack8(hv_ld16(Ksm + kf * C + h * D + dc * 8), k8);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[kf] += q8[e] * k8[e];
            }
        }
    }
    const float c2 = p.scale * 1.44269504089f;
    float mx = -INFINITY;
#pragma unroll
    for (int kf = 0; kf < FMAX; ++kf)
        if (kf < F) mx = fmaxf(mx, s[kf]);
    float l = 0.f;
#pragma unroll
    for (int kf = 0; kf < FMAX; ++kf) {
        s[kf] = kf < F ? __builtin_amdgcn_exp2f((s[kf] - mx) * c2) : 0.f;
        l += s[kf];
    }
    const float inv = 1.0f / l;
    bf16_t* orow = p.O + (row0 + (long)fq * p.P) * p.ldo + h * D;
#pragma unroll
    for (int dc = 0; dc < D / 8; ++dc) {
        float o8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = 0.f;
#pragma unroll
        for (int kf = 0; kf < FMAX; ++kf) {
            if (kf < F) {
                float v8[8];
                hv_unpack8(hv_ld16(Vsm + kf * C + h * D + dc * 8), v8);
#pragma unroll
                for (int e = 0; e < 8; ++e) o8[e] += s[kf] * v8[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] *= inv;
        hv_st16(orow + dc * 8, hv_pack8(o8));
    }
}

template <int D>
static inline int hv_temporal_launch_d(const hv_temporal_attention_params& p, hipStream_t stream) {
    const int threads = ((8 * p.Fq + 63) / 64) * 64;
    const dim3 grid(p.B * p.P), block(threads);
    if (p.Fkv <= 8)
        hv_launch(hv_temporal_kernel<D, 8>, grid, block, stream, p);
    else if (p.Fkv <= 16)
        hv_launch(hv_temporal_kernel<D, 16>, grid, block, stream, p);
    else if (p.Fkv <= 24)
        hv_launch(hv_temporal_kernel<D, 24>, grid, block, stream, p);
    else if (p.Fkv <= 32)
        hv_launch(hv_temporal_kernel<D, 32>, grid, block, stream, p);
    else
        return -2;
    return 0;
}

static inline int hv_temporal_launch(const hv_temporal_attention_params& p, hipStream_t stream) {
    if (p.heads != 8 || p.B <= 0 || p.Fkv <= 0 || p.Fq <= 0 || p.Fq > p.Fkv || p.P <= 0 || p.kv_chunk <= 0)
        return p.heads != 8 ? -2 : -1;
    if (p.ldq % 8 || p.ldkv % 8 || p.ldo % 8) return -1;
    switch (p.D) {
        case 40: return hv_temporal_launch_d<40>(p, stream);
        case 80: return hv_temporal_launch_d<80>(p, stream);
        case 160: return hv_temporal_launch_d<160>(p, stream);
        default: return -2;
    }
}
