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
// of the token-major [(b f) p][3C] tensor.
#pragma once
#include "hv_common.h"
#include "humanvid_hip.h"

// ---- MFMA kernel --------------------------------------------------------------------------------------------------------
// (The round-1 VALU kernel -- one thread per (head, query frame), ~4000 unpack / FMA instructions each -- measured 0.32 ms
// at level 0 against a 0.14 ms HBM roofline and was deleted in round 3.)  Here one WAVE owns one (batch,
// pixel, head): S^T = K.Q^T (keys x queries, at most 32 x 32) and O^T = V^T.P^T are 16x16x32 bf16 MFMAs exactly as in the
// spatial kernel (lane = query column, so the softmax state, P^T and the O^T column live in one lane; the MFMA k-index of
// the P.V product enumerates keys as {4q..4q+3, 16+4q..16+4q+3} per quad q, which is the order S^T leaves them in).
// One workgroup = one (batch, pixel, head group): its Fq query rows and F key / value rows (320 channels = 640 contiguous
// bytes each) are staged row-major in LDS with coalesced 16-byte loads; Q and K fragments are one ds_read_b128 each, a
// V^T fragment is eight 2-byte reads (lanes of a quad read adjacent channels of one key row: conflict-free; transposing
// V while staging it instead put 40 lanes on one bank per store).
template <int D, int FR = 32>
struct HvTemporalGeom {
    static constexpr int HG = 320 / D;                // heads per workgroup (one wave each): 8 / 4 / 2
    static constexpr int CB = HG * D;                 // channels per workgroup (320)
    static constexpr int RS = CB * 2 + 16;            // Q / K row stride in LDS (bytes): odd multiple of 16
    static constexpr int NFULL = D / 32;
    static constexpr bool TAIL = (D % 32) != 0;
    static constexpr int DT = (D + 15) / 16;
    static constexpr int LDS_BYTES = 3 * FR * RS;      // FR rows each of Q, K, V (FR = 24 for windows of <= 24 frames: 47 KB,
                                                       // three workgroups per CU instead of two -- more rows in flight for an HBM-bound op)
};

// Mixed-shape MFMA chains: a 16-deep v_mfma_f32_16x16x16_bf16 chained to 32-deep v_mfma_f32_16x16x32_bf16 steps on the same
// accumulator reads a partially written accumulator on gfx950 with hipcc / ROCm 7.2 (root-caused on the hardware in round 2:
// profiles/r02_mfma_chain_hazard.md).  Rule for every kernel of the library: all MFMAs of one accumulation chain have the same
// shape -- the head-dim remainder (d = 40: 8, d = 80: 16 channels) enters as a zero-padded 32-deep step.
template <int D, int FR>
__global__ __launch_bounds__((HvTemporalGeom<D, FR>::HG * 64)) void hv_temporal_mfma_kernel(hv_temporal_attention_params p) {
    using G = HvTemporalGeom<D, FR>;
    constexpr int HG = G::HG, CB = G::CB, RS = G::RS, NFULL = G::NFULL, DT = G::DT;
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::LDS_BYTES];
    unsigned char* Qs = smem;                // [FR][RS]
    unsigned char* Ks = smem + FR * RS;      // [FR][RS]
    unsigned char* Vs = smem + 2 * FR * RS;  // [FR][RS]
    const int tid = threadIdx.x, nthr = HG * 64;
    const int lane = tid & 63, wave = tid >> 6, r16 = lane & 15, quad = lane >> 4;
    const int F = p.Fkv, FQ = p.Fq;
    const int groups = 8 / HG;
    int t = blockIdx.x;
    const int hg = t % groups;
    t /= groups;
    const int pix = t % p.P, b = t / p.P;
    const int c0 = hg * CB;                              // first channel of this head group
    const long row0 = ((long)b * FQ) * p.P + pix;        // query / output row of local frame f: row0 + f * P

    // ---- stage Q, K, V rows ----
    constexpr int CV = CB / 8;
    auto qo_row = [&](int f) -> long {  // query / output row of local frame f
        return p.qo_chunked ? (long)b * p.kv_stride_b + (long)(f / p.kv_chunk) * p.kv_stride_chunk + (long)(f % p.kv_chunk) * p.P + pix
                            : row0 + (long)f * p.P;
    };
    // every load of the workgroup's Q / K / V rows is issued before the first LDS store (compile-time trip count, predicated):
    // as runtime-trip-count loops hipcc emitted load -> vmcnt(0) -> ds_write per trip, i.e. four dependent HBM round trips
    // per workgroup with one or two 16-byte loads in flight per thread (3.7 TB/s at level 0)
    constexpr int NIT = (FR * CV + HG * 64 - 1) / (HG * 64);
    u32x4 qr[NIT], kr[NIT], vr[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {  // unconditional loads from clamped (always valid) rows: no branch, no wait between them
        const int i = tid + it * nthr;
        const int c = i % CV;
        const int fq = min(i / CV, FQ - 1), fk = min(i / CV, F - 1);
        const long kvrow = (long)b * p.kv_stride_b + (long)(fk / p.kv_chunk) * p.kv_stride_chunk + (long)(fk % p.kv_chunk) * p.P + pix;
        qr[it] = hv_ld16(p.Q + qo_row(fq) * p.ldq + c0 + c * 8);
        kr[it] = hv_ld16(p.K + kvrow * p.ldkv + c0 + c * 8);
        vr[it] = hv_ld16(p.V + kvrow * p.ldkv + c0 + c * 8);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + it * nthr;
        const int f = i / CV, c = i % CV;
        if (i < FQ * CV) hv_st16(Qs + f * RS + c * 16, qr[it]);
        if (i < F * CV) {
            hv_st16(Ks + f * RS + c * 16, kr[it]);
            hv_st16(Vs + f * RS + c * 16, vr[it]);
        }
    }
    __syncthreads();

    // ---- this wave's head ----
    const int hd = wave * D;  // channel offset of the head inside the group
    const int nqt = (FQ + 15) >> 4, nkt = (F + 15) >> 4;
    const float c2 = p.scale * 1.44269504089f;
    f32x4 sacc[2][2];  // [key tile][query tile]: lane = (query r16, quad), reg r <-> key 16*kt + 4*quad + r
    // remainder of the head dim past the 32-deep steps (8 channels for d = 40, 16 for d = 80)
    auto acc_tail = [&](f32x4& acc, const unsigned char* krow, const unsigned char* qrow) {
        // zero-padded 32-deep step: quads whose 8 channels lie past D contribute zeros (same MFMA shape as the rest of the chain)
        u32x4 ka = {0u, 0u, 0u, 0u}, qa = {0u, 0u, 0u, 0u};
        if (32 * NFULL + 8 * quad + 8 <= D) {
            ka = hv_ld16(krow + NFULL * 64 + quad * 16);
            qa = hv_ld16(qrow + NFULL * 64 + quad * 16);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hv_as_bf16x8(ka), hv_as_bf16x8(qa), acc, 0, 0, 0);
    };
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) sacc[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        if (kt >= nkt) break;
        const unsigned char* krow = Ks + min(16 * kt + r16, F - 1) * RS + hd * 2;  // rows past F: clamped, masked below
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            if (qt >= nqt) break;
            const unsigned char* qrow = Qs + min(16 * qt + r16, FQ - 1) * RS + hd * 2;
#pragma unroll
            for (int s = 0; s < NFULL; ++s)
            {
                sacc[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hv_as_bf16x8(hv_ld16(krow + s * 64 + quad * 16)),
                                                                       hv_as_bf16x8(hv_ld16(qrow + s * 64 + quad * 16)),
                                                                       sacc[kt][qt], 0, 0, 0);
            }
            if (G::TAIL) acc_tail(sacc[kt][qt], krow, qrow);
        }
    }
    // ---- softmax over the keys of each query column (all keys are here: no running state) ----
    bf16x8 pf[2];
    float inv_l[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (16 * kt + 4 * quad + r >= F) sacc[kt][qt][r] = -INFINITY;
                mx = fmaxf(mx, sacc[kt][qt][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float pv[2][4], l = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pv[kt][r] = __builtin_amdgcn_exp2f((sacc[kt][qt][r] - mx) * c2);
                l += pv[kt][r];
            }
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        inv_l[qt] = 1.0f / l;
        const u32x4 w = {hv_pack2(pv[0][0], pv[0][1]), hv_pack2(pv[0][2], pv[0][3]), hv_pack2(pv[1][0], pv[1][1]),
                         hv_pack2(pv[1][2], pv[1][3])};
        pf[qt] = hv_as_bf16x8(w);
    }
    // ---- O^T = V^T . P^T, normalise, store: lane owns query r16 (+16 qt), channels 16 dt + 4 quad + 0..3 ----
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        // V^T fragment: lane (channel 16 dt + r16, quad) holds keys {4 quad .. 4 quad + 3, 16 + 4 quad .. 16 + 4 quad + 3};
        // keys past F meet P = 0 and channels past D (d = 40: fragment 2) are never stored: clamp both to valid data
        const unsigned char* vcol = Vs + (hd + min(16 * dt + r16, D - 1)) * 2;
        unsigned vw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k0 = 16 * (j >> 1) + 4 * quad + 2 * (j & 1);
            const unsigned lo = *reinterpret_cast<const bf16_t*>(vcol + min(k0, F - 1) * RS);
            const unsigned hi = *reinterpret_cast<const bf16_t*>(vcol + min(k0 + 1, F - 1) * RS);
            vw[j] = lo | (hi << 16);
        }
        const bf16x8 vf = hv_as_bf16x8(u32x4{vw[0], vw[1], vw[2], vw[3]});
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            if (qt >= nqt) break;
            f32x4 o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qt], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            const int q = 16 * qt + r16, d = 16 * dt + 4 * quad;
            if (q < FQ && d < D) {
                const u32x2 st = {hv_pack2(o[0] * inv_l[qt], o[1] * inv_l[qt]), hv_pack2(o[2] * inv_l[qt], o[3] * inv_l[qt])};
                hv_st8(p.O + qo_row(q) * p.ldo + c0 + hd + d, st);
            }
        }
    }
}

static inline int hv_temporal_launch(const hv_temporal_attention_params& p, hipStream_t stream) {
    if (p.heads != 8 || p.B <= 0 || p.Fkv <= 0 || p.Fq <= 0 || p.Fq > p.Fkv || p.P <= 0 || p.kv_chunk <= 0)
        return p.heads != 8 ? -2 : -1;
    if (p.ldq % 8 || p.ldkv % 8 || p.ldo % 8) return -1;
    if (p.qo_chunked && p.Fq != p.Fkv) return -1;
    if (p.Fkv > 32) return -2;  // one 32-key MFMA tile per (batch, pixel, head): the positional encoding caps windows at 32 frames
    hv_note("hv_temporal_mfma_kernel<%d> | B=%d Fq=%d Fkv=%d P=%d", p.D, p.B, p.Fq, p.Fkv, p.P);
    const bool fr24 = p.Fkv <= 24;
    switch (p.D) {
        case 40:
            if (fr24) hv_launch(hv_temporal_mfma_kernel<40, 24>, dim3(p.B * p.P), dim3(512), stream, p);
            else hv_launch(hv_temporal_mfma_kernel<40, 32>, dim3(p.B * p.P), dim3(512), stream, p);
            return 0;
        case 80:
            if (fr24) hv_launch(hv_temporal_mfma_kernel<80, 24>, dim3(p.B * p.P * 2), dim3(256), stream, p);
            else hv_launch(hv_temporal_mfma_kernel<80, 32>, dim3(p.B * p.P * 2), dim3(256), stream, p);
            return 0;
        case 160:
            if (fr24) hv_launch(hv_temporal_mfma_kernel<160, 24>, dim3(p.B * p.P * 4), dim3(128), stream, p);
            else hv_launch(hv_temporal_mfma_kernel<160, 32>, dim3(p.B * p.P * 4), dim3(128), stream, p);
            return 0;
        default: return -2;
    }
}
