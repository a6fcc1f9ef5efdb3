// hv_attention.h -- flash-style spatial self-attention with reference-bank keys on MFMA (gfx950).
//
// Reference semantics: diffusers Attention + AttnProcessor2_0 (F.scaled_dot_product_attention,
// scale d^-0.5, no mask) as called by the read-mode patched transformer block,
// /root/reference/src/models/mutual_self_attention.py:147-186: for the conditional images keys and
// values are [own tokens || bank tokens]; for the CFG-unconditional images the result is
// overwritten by plain self-attention, i.e. they attend to their own tokens only.  This kernel
// computes exactly that "overwritten" result, selecting the bank per image.
//
// Design (MI355X, head dims 40 / 80 / 160 = SD-1.5 widths 320/640/1280 over 8 heads):
//  * S^T = K.Q^T with mfma_f32_16x16x32_bf16 (A = key rows, B = query rows): a lane then holds
//    scores of ONE query (column lane&15) for 4 keys per fragment, so the softmax row reduction is
//    16 in-register values + two cross-quad shuffles, and the running max / rescale factor of a query
//    live in the same lane as its O^T accumulator column (no broadcast needed).
//  * the head dim is not a multiple of 32 for d = 40 / 80: the remainder of the QK^T reduction uses
//    the 16-deep mfma_f32_16x16x16_bf16 so d = 40 costs 48 (not 64) and d = 80 costs exactly 80.
//  * key rows are staged into LDS in a permuted order (bits 2 and 3-4 rotated) so that the
//    probabilities of two adjacent score fragments concatenate, in-register, into the B operand of
//    O^T += V^T.P^T with natural key order -- no cross-lane traffic between the two matmuls.
//  * V arrives already transposed ([channel][token], written by the QKV GEMM epilogue), so both
//    K and V^T fragments are single LDS vector reads; LDS rows are padded to an odd number of
//    16-byte slots (conflict-free ds_read_b128).
//  * the O^T rescale is skipped for tiles in which no query of the wave raised its running maximum
//    (wave-uniform test); the ragged-tail masking is a separate template instance.
//  * K/V tiles (64 keys) are double-buffered through registers, one barrier per tile; online
//    softmax in the exp2 domain; fp32 accumulation; register budget sized for >= 2 workgroups per CU.
#pragma once
#include "hv_common.h"
#include "humanvid_hip.h"

// phase timestamps for tools/attn_trace.hip (which includes hv_gemm.h first with HV_GEMM_TRACE defined)
#ifndef HV_TRACE
#define HV_TRACE(id)
#endif

#ifndef HV_ATTN_OCC40
#define HV_ATTN_OCC40 4
#endif
#ifndef HV_ATTN_ONES
// Denominator through the MFMA: d = 40 pads V^T to 48 rows; the first spare row is a row of ones, so the P.V MFMA that is
// issued anyway also accumulates sum(P) and the 16 VALU adds per query fragment and tile disappear (level 0: 6.21 -> 5.89 ms
// same-box).  Round 1 found this variant "wrong on hardware" and parked it; that was the mixed-shape MFMA chain of the
// QK^T remainder (see HV_ATTN_PAD32 below; with fewer VALU instructions between the dependent pair it tripped more
// often).  With same-shape chains it passes every hardware case, bench shapes included.
#define HV_ATTN_ONES 1
#endif
#ifndef HV_ATTN_LAZY
#define HV_ATTN_LAZY 1
#endif
#ifndef HV_ATTN_DEFER
// Softmax with fewer VALU instructions per score (round 2).  At d = 40 the softmax is the longest phase of a tile: per
// score one max, one fused multiply-subtract, one v_exp_f32 (half rate) and half a packed convert -- ~750 VALU cycles per
// 64-key tile against 448 MFMA cycles.  Here (a) the queries are pre-multiplied by scale * log2(e) once at load, (b) the
// running reference maximum enters the QK^T MFMA as the accumulator's initial value (C = -m), so the MFMA delivers
// s - m directly, and (c) the reference maximum is only raised when a query's tile maximum exceeds it by more than
// HV_ATTN_THR (in log2 units; the probabilities are then bounded by 2^THR instead of 1, harmless in bf16 x fp32) -- the
// common tile needs max + exp2 + convert per score and no multiply-subtract.  The rescale branch (wave-uniform) subtracts
// the increase from the scores BEFORE they are exponentiated and scales O (and the denominator row) by 2^-increase, i.e.
// everything still at the old reference is scaled exactly once.  0 = the round-1 softmax, for A/Bs.
#define HV_ATTN_DEFER 1
#endif
#ifndef HV_ATTN_THR
#define HV_ATTN_THR 8.0f
#endif
#ifndef HV_ATTN_LOCALMAX
#define HV_ATTN_LOCALMAX 1  // round 3: lane-local v_max3 maxima on the common path (0 = the round-2 reduction, for A/Bs)
#endif

// Head-dim remainder of the QK^T reduction (d = 40: 8 channels, d = 80: 16).  Round 1 fed it through a 16-deep
// mfma_f32_16x16x16_bf16 appended to the chain of 32-deep MFMAs on the same accumulator.  That mix is unsafe on gfx950
// with hipcc (ROCm 7.2): a v_mfma_f32_16x16x16_bf16 that takes the result of a v_mfma_f32_16x16x32_bf16 as SrcC (or the
// reverse) is issued back to back, without the wait states / independent instructions the different pass counts need,
// and reads a partially written accumulator -- wrong rows that change run to run, depending on what else shares the
// SIMD (root-caused on the temporal kernel: tools/diag_fence.py, profiles/r02_mfma_chain_hazard.md).  This kernel never
// showed it (its two query fragments interleave independent MFMAs between the dependent pair), but nothing guaranteed
// that.  HV_ATTN_PAD32 = 1 (default) keeps every MFMA of a chain the same shape: the remainder becomes a zero-padded
// 32-deep step (K rows in LDS are zero beyond D, the query fragment is masked).  0 = the round-1 form, for A/Bs only.
#ifndef HV_ATTN_PAD32
#define HV_ATTN_PAD32 1
#endif

template <int D, int QT>
struct HvAttnGeom {
    static constexpr int NFULL = HV_ATTN_PAD32 ? (D + 31) / 32 : D / 32;   // 32-deep QK^T steps
    static constexpr bool TAIL = !HV_ATTN_PAD32 && (D % 32) != 0;          // + one 16-deep step (round-1 form)
    static constexpr int DT = (D + 15) / 16;          // 16-row fragments of V^T / O^T
    static constexpr int DK = 32 * NFULL + (TAIL ? 16 : 0);
    static constexpr int DV = 16 * DT;
    static constexpr bool ONES = HV_ATTN_ONES && DV > D;              // spare V^T row available for the denominator
    static constexpr int KRS = DK * 2 + 16;           // K row stride in LDS (bytes), odd multiple of 16
    static constexpr int VRS = 64 * 2 + 16;           // V^T row stride (64 keys)
    static constexpr int KBYTES = 64 * KRS;
    static constexpr int VBYTES = DV * VRS;
    static constexpr int KCH = 64 * (D / 8);          // 16-byte chunks of a K tile
    static constexpr int VCH = D * 8;                 // 16-byte chunks of a V^T tile
    static constexpr int KIT = (KCH + 255) / 256;
    static constexpr int VIT = (VCH + 255) / 256;
    static constexpr int BQ = 4 * 16 * QT;            // queries per workgroup
};

// waves per SIMD the register allocator is asked to fit (LDS allows 5 / 3 / 1 workgroups per CU for
// d = 40 / 80 / 160): more resident waves let one wave's softmax VALU overlap another's MFMA
template <int D, int QT>
struct HvAttnOcc {
    static constexpr int value = (D == 40 && QT == 2) ? HV_ATTN_OCC40 : ((D == 80 && QT == 2) ? 2 : 1);
};

template <int D, int QT, bool MASK>
__global__ __launch_bounds__(256, (HvAttnOcc<D, QT>::value)) void hv_attention_kernel(hv_attention_params p) {
    using G = HvAttnGeom<D, QT>;
    constexpr int NFULL = G::NFULL, DT = G::DT;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (G::KBYTES + G::VBYTES)];
    unsigned char* Ks = smem;
    unsigned char* Vs = smem + 2 * G::KBYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, quad = lane >> 4;

    const int nqb = (p.Lq + G::BQ - 1) / G::BQ;
    const int total = nqb * p.heads * p.n_images;
    const int cpx = gridDim.x / 8;
    int t = (blockIdx.x % 8) * cpx + blockIdx.x / 8;
    if (t >= total) return;
    const int qb = t % nqb;
    t /= nqb;
    const int head = t % p.heads;
    int img = t / p.heads;
    // XCD x walks the contiguous range [x cpx, (x + 1) cpx) of t.  With the images in storage order the first four XCDs got the
    // CFG-unconditional half (own keys only) and the last four the conditional half (own + bank keys = twice the work): the
    // launch lasted as long as the heavy half.  Alternating between the halves gives every XCD the same mix.
    if ((p.n_images & 1) == 0) img = (img & 1) * (p.n_images >> 1) + (img >> 1);
    const int sel = (p.bank_sel != nullptr && p.L2 > 0) ? p.bank_sel[img] : -1;
    const int T1 = (p.L1 + 63) / 64;
    const int T2 = sel >= 0 ? (p.L2 + 63) / 64 : 0;
    const int ntiles = T1 + T2;

    // zero the padding of both LDS buffers once (never overwritten by the tile stores); the first
    // spare V^T row becomes a row of ones (bf16 1.0 = 0x3F80) -> denominator via the P.V MFMA
    for (int i = tid; i < 2 * (G::KBYTES + G::VBYTES) / 16; i += 256) {
        u32x4 z = {0u, 0u, 0u, 0u};
        hv_st16(smem + i * 16, z);
    }
    __syncthreads();
    if (G::ONES) {
        for (int i = tid; i < 2 * 8; i += 256) {
            u32x4 one = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
            hv_st16(Vs + (i >> 3) * G::VBYTES + D * G::VRS + (i & 7) * 16, one);
        }
    }

    // ---- query fragments (B operand of S^T = K.Q^T), resident for the whole kernel
    bf16x8 qf[QT][NFULL > 0 ? NFULL : 1];
    bf16x4 qtail[QT];
    const int q_wave = qb * G::BQ + wave * 16 * QT;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int q = q_wave + 16 * qt + r16;
        const bf16_t* qrow = p.Q + ((long)img * p.Lq + q) * p.ldq + head * D;
#pragma unroll
        for (int s = 0; s < NFULL; ++s) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (q < p.Lq && 32 * s + 8 * quad + 8 <= D) v = hv_ld16(qrow + 32 * s + 8 * quad);
            if (HV_ATTN_DEFER) {  // scores come out of the MFMA in the exp2 domain
                float f[8];
                hv_unpack8(v, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] *= p.scale * 1.44269504089f;
                v = hv_pack8(f);
            }
            qf[qt][s] = hv_as_bf16x8(v);
        }
        if (G::TAIL) {
            const int d = 32 * NFULL + 4 * quad;
            u32x2 v = {0u, 0u};
            if (q < p.Lq && d + 4 <= D) v = hv_ld8(qrow + d);
            union {
                u32x2 u;
                bf16x4 s;
            } c;
            c.u = v;
            qtail[qt] = c.s;
        }
    }

    u32x4 kreg[G::KIT], vreg[G::VIT];
    // Per-lane parts of the K / V^T tile addresses, computed once (32-bit byte offsets from the wave-uniform tensor base:
    // hv_attention_launch checks the spans); per tile only a scalar offset changes.  (Round 1 rebuilt every 64-bit address
    // from scratch in every tile: ~60 of the ~290 instructions of a tile.)
    unsigned koff[G::KIT], voff[G::VIT], koff2[G::KIT], voff2[G::VIT];
#pragma unroll
    for (int i = 0; i < G::KIT; ++i) {
        const int id = tid + 256 * i;
        const int r = id / (D / 8), c = id % (D / 8);
        koff[i] = ((unsigned)r * (unsigned)p.ldk + (unsigned)c * 8u) * 2u;
        koff2[i] = ((unsigned)r * (unsigned)p.ldk2 + (unsigned)c * 8u) * 2u;
    }
#pragma unroll
    for (int i = 0; i < G::VIT; ++i) {
        const int id = tid + 256 * i;
        voff[i] = ((unsigned)(id >> 3) * (unsigned)p.ldvt + (unsigned)(id & 7) * 8u) * 2u;
        voff2[i] = ((unsigned)(id >> 3) * (unsigned)p.ldvt2 + (unsigned)(id & 7) * 8u) * 2u;
    }
    auto load_tile = [&](int ti) {
        const bool bank = ti >= T1;
        const int kv0 = (bank ? ti - T1 : ti) * 64;
        const int L = bank ? p.L2 : p.L1;
        const unsigned rowbase = bank ? (unsigned)sel * (unsigned)p.L2 : (unsigned)img * (unsigned)p.L1;
        const char* Kp = reinterpret_cast<const char*>(bank ? p.K2 : p.K);
        const char* Vp = reinterpret_cast<const char*>(bank ? p.Vt2 : p.Vt);
        // wave-uniform parts
        const unsigned kb = ((rowbase + (unsigned)kv0) * (unsigned)(bank ? p.ldk2 : p.ldk) + (unsigned)(head * D)) * 2u;
        const unsigned vb = ((unsigned)(head * D) * (unsigned)(bank ? p.ldvt2 : p.ldvt) + rowbase + (unsigned)kv0) * 2u;
#pragma unroll
        for (int i = 0; i < G::KIT; ++i) {
            const int id = tid + 256 * i;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (id < G::KCH) {
                const int r = id / (D / 8);
                if (!MASK || kv0 + r < L) v = hv_ld16(Kp + (kb + (bank ? koff2[i] : koff[i])));
            }
            kreg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < G::VIT; ++i) {
            const int id = tid + 256 * i;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (id < G::VCH) {
                const int c = id & 7;
                if (!MASK || kv0 + c * 8 < L) v = hv_ld16(Vp + (vb + (bank ? voff2[i] : voff[i])));
            }
            vreg[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < G::KIT; ++i) {
            const int id = tid + 256 * i;
            if (id < G::KCH) {
                const int r = id / (D / 8), c = id % (D / 8);
                // key kv = 32a + 8b + 4c' + e  ->  LDS row 32a + 16c' + 4b + e
                const int lr = (r & ~0x1c) | ((r & 4) << 2) | ((r & 0x18) >> 1);
                hv_st16(Ks + buf * G::KBYTES + lr * G::KRS + c * 16, kreg[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < G::VIT; ++i) {
            const int id = tid + 256 * i;
            if (id < G::VCH) hv_st16(Vs + buf * G::VBYTES + (id >> 3) * G::VRS + (id & 7) * 16, vreg[i]);
        }
    };

    f32x4 oacc[QT][DT];
    float mrun[QT], lrun[QT];
    f32x4 cneg[QT];  // HV_ATTN_DEFER: the query's reference maximum, negated, as the QK^T accumulator's initial value (changes
                     // only in the rescale branch: kept as a register quad instead of being rebuilt in every tile)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        mrun[qt] = HV_ATTN_DEFER ? 0.f : -INFINITY;
        cneg[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
        lrun[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) oacc[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float c2 = p.scale * 1.44269504089f;

#ifdef HV_GEMM_TRACE
    int hv_ti = 0;
#endif
    load_tile(0);
    __syncthreads();  // LDS initialisation complete before the first tile store
    for (int ti = 0; ti < ntiles; ++ti) {
        const int buf = ti & 1;
        HV_TRACE(1);
        store_tile(buf);
        HV_TRACE(2);
        __syncthreads();
        HV_TRACE(3);
        if (ti + 1 < ntiles) load_tile(ti + 1);
        HV_TRACE(4);
        const unsigned char* kb = Ks + buf * G::KBYTES + r16 * G::KRS;
        const unsigned char* vb = Vs + buf * G::VBYTES + r16 * G::VRS + quad * 16;

        // ---- S^T fragments: sacc[kvf][qt], lane = (query r16, quad), reg r <-> key
        //      kv = 32*(kvf>>1) + 8*quad + 4*(kvf&1) + r
        f32x4 sacc[4][QT];
#pragma unroll
        for (int kvf = 0; kvf < 4; ++kvf) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) sacc[kvf][qt] = cneg[qt];  // lane = one query: -(its reference maximum), or 0
#pragma unroll
            for (int s = 0; s < NFULL; ++s) {
                const bf16x8 kf = hv_as_bf16x8(hv_ld16(kb + (16 * kvf) * G::KRS + s * 64 + quad * 16));
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    sacc[kvf][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][s], sacc[kvf][qt], 0, 0, 0);
            }
            if (G::TAIL) {
                union {
                    u32x2 u;
                    bf16x4 s;
                } kt;
                kt.u = hv_ld8(kb + (16 * kvf) * G::KRS + NFULL * 64 + quad * 8);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    sacc[kvf][qt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kt.s, qtail[qt], sacc[kvf][qt], 0, 0, 0);
            }
        }
        // ---- mask the ragged tail of this source (MASK instances only)
        if (MASK) {
            const bool bank = ti >= T1;
            const int kv0 = (bank ? ti - T1 : ti) * 64;
            const int L = bank ? p.L2 : p.L1;
            if (kv0 + 64 > L) {
#pragma unroll
                for (int kvf = 0; kvf < 4; ++kvf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kv = kv0 + 32 * (kvf >> 1) + 8 * quad + 4 * (kvf & 1) + r;
                        if (kv >= L) {
#pragma unroll
                            for (int qt = 0; qt < QT; ++qt) sacc[kvf][qt][r] = -INFINITY;
                        }
                    }
            }
        }
        HV_TRACE(5);
        // ---- online softmax (exp2 domain) and P^T fragments
        bf16x8 pf[QT][2];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
#if HV_ATTN_DEFER && HV_ATTN_LOCALMAX
            // lane-local maximum of the lane's 16 scores: eight v_max3_f32 (hipcc's fmaxf canonicalises every operand
            // first -- a v_max x, x in front of each real maximum: 28 instructions for these 16 values).  The cross-quad
            // reduction to the query's tile maximum (two ds_bpermute round trips) moves into the rare branch: "some query
            // of this wave exceeds the threshold" is the same predicate over lane-local maxima as over reduced ones.
            float mx = hv_max3(sacc[0][qt][0], sacc[0][qt][1], sacc[0][qt][2]);
            mx = hv_max3(mx, sacc[0][qt][3], sacc[1][qt][0]);
            mx = hv_max3(mx, sacc[1][qt][1], sacc[1][qt][2]);
            mx = hv_max3(mx, sacc[1][qt][3], sacc[2][qt][0]);
            mx = hv_max3(mx, sacc[2][qt][1], sacc[2][qt][2]);
            mx = hv_max3(mx, sacc[2][qt][3], sacc[3][qt][0]);
            mx = hv_max3(mx, sacc[3][qt][1], sacc[3][qt][2]);
            mx = hv_max3(mx, sacc[3][qt][3], sacc[3][qt][3]);
#else
            float mx = fmaxf(fmaxf(sacc[0][qt][0], sacc[0][qt][1]), fmaxf(sacc[0][qt][2], sacc[0][qt][3]));
#pragma unroll
            for (int kvf = 1; kvf < 4; ++kvf)
                mx = fmaxf(fmaxf(fmaxf(mx, sacc[kvf][qt][0]), fmaxf(sacc[kvf][qt][1], sacc[kvf][qt][2])),
                           sacc[kvf][qt][3]);
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
#endif
#if HV_ATTN_DEFER
            static_assert(!G::TAIL, "HV_ATTN_DEFER needs HV_ATTN_PAD32 (pre-scaled queries have no 16-deep tail fragment)");
            float pv[4][4];
            {
                // exponentials first, against the current reference: they do not depend on this tile's maximum unless the
                // (rare) rescale branch fires, so the max reduction and its two cross-lane steps run beside them
                // instead of in front of them
                float psum = 0.f;
#pragma unroll
                for (int kvf = 0; kvf < 4; ++kvf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[kvf][r] = __builtin_amdgcn_exp2f(sacc[kvf][qt][r]);
                const bool first = ti == 0;  // the first tile fixes the reference maximum (it starts at 0, not at a score)
                if (first || __any(mx > HV_ATTN_THR)) {
#if HV_ATTN_LOCALMAX
                    mx = fmaxf(mx, __shfl_xor(mx, 16));
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
#endif
                    const float inc = first ? mx : fmaxf(mx, 0.f);
#pragma unroll
                    for (int kvf = 0; kvf < 4; ++kvf)
#pragma unroll
                        for (int r = 0; r < 4; ++r) pv[kvf][r] = __builtin_amdgcn_exp2f(sacc[kvf][qt][r] - inc);
                    if (!first) {
                        const float alpha = __builtin_amdgcn_exp2f(-inc);
                        if (!G::ONES) lrun[qt] *= alpha;
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) oacc[qt][dt] *= alpha;
                    }
                    mrun[qt] += inc;
                    cneg[qt] -= f32x4{inc, inc, inc, inc};
                }
                if (!G::ONES) {
#pragma unroll
                    for (int kvf = 0; kvf < 4; ++kvf)
#pragma unroll
                        for (int r = 0; r < 4; ++r) psum += pv[kvf][r];
                    lrun[qt] += psum;
                }
            }
#else
            const float mold = mrun[qt];
            const float mnew = fmaxf(mold, mx * c2);
            mrun[qt] = mnew;
            float pv[4][4];
            float psum = 0.f;
#pragma unroll
            for (int kvf = 0; kvf < 4; ++kvf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pv[kvf][r] = __builtin_amdgcn_exp2f(sacc[kvf][qt][r] * c2 - mnew);
                    if (!G::ONES) psum += pv[kvf][r];
                }
            if (!HV_ATTN_LAZY || __any(mnew > mold)) {  // some query of this wave raised its maximum: rescale the accumulators
                const float alpha = __builtin_amdgcn_exp2f(mold - mnew);
                if (!G::ONES) lrun[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) oacc[qt][dt] *= alpha;
            }
            if (!G::ONES) lrun[qt] += psum;
#endif
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 w = {hv_pack2(pv[2 * ks][0], pv[2 * ks][1]), hv_pack2(pv[2 * ks][2], pv[2 * ks][3]),
                           hv_pack2(pv[2 * ks + 1][0], pv[2 * ks + 1][1]),
                           hv_pack2(pv[2 * ks + 1][2], pv[2 * ks + 1][3])};
                pf[qt][ks] = hv_as_bf16x8(w);
            }
        }
        HV_TRACE(6);
        // ---- O^T += V^T . P^T   (row D of V^T is all ones when ONES: accumulates the denominator)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 vf = hv_as_bf16x8(hv_ld16(vb + (16 * dt) * G::VRS + ks * 64));
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    oacc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qt][ks], oacc[qt][dt], 0, 0, 0);
            }
        HV_TRACE(7);
    }

    // ---- normalise and store: lane owns query r16, channels 16*dt + 4*quad + 0..3
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l;
        if (G::ONES) {
            // denominator = O^T row D: fragment D/16, row (D%16) = 4*quad' + r  ->  lane quad' = (D%16)/4, reg (D%4)
            l = __shfl(oacc[qt][D / 16][D % 4], ((D % 16) / 4) * 16 + r16);
        } else {
            l = lrun[qt];
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
        }
        const float inv = 1.0f / l;
        const int q = q_wave + 16 * qt + r16;
        if (q >= p.Lq) continue;
        bf16_t* dst = p.O + ((long)img * p.Lq + q) * p.ldo + head * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = 16 * dt + 4 * quad;
            if (d < D) {
                u32x2 o = {hv_pack2(oacc[qt][dt][0] * inv, oacc[qt][dt][1] * inv),
                           hv_pack2(oacc[qt][dt][2] * inv, oacc[qt][dt][3] * inv)};
                hv_st8(dst + d, o);
            }
        }
    }
}

template <int D, int QT>
static inline void hv_attention_launch_t(const hv_attention_params& p, hipStream_t stream) {
    using G = HvAttnGeom<D, QT>;
    const int total = ((p.Lq + G::BQ - 1) / G::BQ) * p.heads * p.n_images;
    const int grid = ((total + 7) / 8) * 8;
    const bool ragged = (p.L1 % 64) != 0 || (p.L2 % 64) != 0;
    hv_note("hv_attention_kernel<%d,%d> | n=%d heads=%d D=%d Lq=%d L1=%d L2=%d bank=%d", D, QT, p.n_images, p.heads, D, p.Lq,
            p.L1, p.L2, p.bank_sel != nullptr && p.L2 > 0);
    if (ragged)
        hv_launch(hv_attention_kernel<D, QT, true>, dim3(grid), dim3(256), stream, p);
    else
        hv_launch(hv_attention_kernel<D, QT, false>, dim3(grid), dim3(256), stream, p);
}

// tuning knobs (hv_set_tuning): query fragments per wave, per head dim
static int g_hv_attn_qt40 = 2, g_hv_attn_qt160 = 2;

static inline int hv_attention_launch(const hv_attention_params& p, hipStream_t stream) {
    if (p.L1 <= 0 || p.L1 % 8 != 0 || p.L2 % 8 != 0 || p.Lq <= 0) return -1;
    if (p.ldq % 8 || p.ldk % 8 || p.ldvt % 8 || p.ldo % 4) return -1;
    if (p.L2 > 0 && p.bank_sel != nullptr && (!p.K2 || !p.Vt2 || p.ldk2 % 8 || p.ldvt2 % 8)) return -1;
    {  // 32-bit byte offsets inside the kernel
        const long lim = 1L << 32, C = (long)p.heads * p.D;
        if ((long)p.n_images * p.L1 * p.ldk * 2 >= lim || (C * p.ldvt + (long)p.n_images * p.L1) * 2 >= lim) return -1;
        if (p.L2 > 0 && p.bank_sel != nullptr && (64L * p.L2 * p.ldk2 * 2 >= lim || (C * p.ldvt2 + 64L * p.L2) * 2 >= lim))
            return -1;
    }
    switch (p.D) {
        case 40:
            if (g_hv_attn_qt40 == 4) hv_attention_launch_t<40, 4>(p, stream);
            else hv_attention_launch_t<40, 2>(p, stream);
            break;
        case 80: hv_attention_launch_t<80, 2>(p, stream); break;
        case 160:
            if (g_hv_attn_qt160 == 1) hv_attention_launch_t<160, 1>(p, stream);
            else hv_attention_launch_t<160, 2>(p, stream);
            break;
        default: return -2;
    }
    return 0;
}
