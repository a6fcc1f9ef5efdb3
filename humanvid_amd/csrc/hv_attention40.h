// hv_attention40.h -- spatial self-attention with reference-bank keys for head dim 40 (SD-1.5 level 0: 320 channels over
// 8 heads), the shape that carries 19 % of a denoising step at 24f x 768x512 (five launches of 48 images x 6144 queries
// against 6144 own + 6144 bank keys).  Same semantics and parameter block as hv_attention (hv_attention.h; reference:
// /root/reference/src/models/mutual_self_attention.py:147-186 over diffusers' AttnProcessor2_0), other structure.
//
// What the round-3 measurements said about the generic kernel at this shape (profiles/r03_attn_ablation.txt, 4.2 ms per
// launch): the MFMAs alone take 1.47 ms -- 40 % of them padding (QK^T 40 -> 64 deep, PV 40 -> 48 rows) --, the 14 fragment
// reads per tile and wave cost 1.5 ms, staging the K / V^T tiles through registers into LDS 1.55 ms, softmax VALU 1.1 ms.
// This kernel (round 4) changes the four things those numbers point at:
//  * S^T = K.Q^T on v_mfma_f32_32x32x16_bf16: 16-deep steps pad the head dim to 48 instead of 64 (-25 % QK^T MFMA time),
//    one K fragment read feeds a 32-cycle MFMA over 32 queries (six K reads per 64-key tile and wave instead of eight), and
//    no accumulation chain mixes MFMA shapes (the gfx950 hazard of profiles/r02_mfma_chain_hazard.md): S^T chains are
//    32x32x16 only, O^T chains 16x16x32 only;
//  * the probabilities go from the 32 x 32 S^T layout (lane = query l & 31, half l >> 5 holds 16 of the block's 32 keys)
//    to the B operand of O^T += V^T.P^T (16x16x32: lane = query l & 15, quad l >> 4 holds 8 keys) with four
//    v_permlane16_swap per 32-key block -- the K rows are staged in the order that makes the swapped dwords line up with
//    the natural key order of V^T (below), so nothing else moves;
//  * 256 queries per workgroup (8 waves x 32): a staged K / V^T tile serves twice the queries of the generic kernel's 128
//    -- half the staging loads, LDS stores and barriers per query; 30 KB of LDS, two workgroups per CU;
//  * softmax VALU work per score: the query's reference maximum rides in the MFMA (K is augmented by a column of ones at
//    head-dim index 40, Q by -m there: the MFMA delivers s - m with C = 0 -- no accumulator initialisation moves, no
//    subtraction), the "some probability exceeds 2^THR" test runs on the PACKED bf16 pairs (v_pk_max_u16: positive bf16
//    order like unsigned integers -- 15 instead of 31 maxima), exponentials in the exp2 domain as before.
//  * OPTIMISTIC reference maximum: the first tile's maximum is kept as the reference for the whole key loop and NO per-tile
//    test runs (the "some probability exceeds 2^THR" test of the generic kernel -- 15 packed maxima, a compare and a
//    wave vote per tile -- measured 6.3 % of the kernel, profiles/r04_s3.txt).  bf16 probabilities and fp32 accumulators
//    keep their relative precision at any magnitude, so nothing is lost until a later score exceeds the reference by ~2^7
//    in the exp2 domain and the exponential overflows; the denominator row of O^T then is not finite, the workgroup votes
//    once after the loop, and -- only then -- runs the loop again in the careful form (test + deferred rescale per tile).
// The reference maximum therefore lives in bf16 (it is an element of the Q operand); the rescale factors are computed from
// the rounded values, so every tile of a query is exponentiated against exactly the maximum its O^T / denominator carry
// (the softmax is invariant to the choice of reference as long as it is used consistently).
//
// Key order.  S^T block (32 keys x 32 queries), C/D layout of the 32x32 MFMA: lane l, register i <-> row (i & 3) + 8 (i >> 2)
// + 4 (l >> 5).  K tile rows are staged so that LDS row 8 a + 4 b + c of a 32-key block holds key 16 b + 4 a + c: register i
// of half b then IS key 16 b + i, the packed pair j (registers 2 j, 2 j + 1) keys 16 b + 2 j (+1), and after
// v_permlane16_swap(pair d, pair d + 4) quad q of the result holds keys 8 q + 2 d (+1) for the queries r16 (first result:
// queries 0-15 of the wave's 32, second: 16-31) -- dword d of the PV B operand with V^T in natural key order.
#pragma once
#include "hv_common.h"
#include "humanvid_hip.h"

#ifndef HV_ATTN_THR
#define HV_ATTN_THR 8.0f
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short hv_u16x2 __attribute__((ext_vector_type(2)));

// v_permlane16_swap: odd 16-lane rows of `a` are exchanged with even rows of `b`
HV_DEV void hv_swap16_pair(unsigned& a, unsigned& b) {
#ifndef HV_EMU
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
#else
    const int lane = threadIdx.x & 63;
    const bool odd = (lane >> 4) & 1;
    const unsigned xa = __shfl(b, lane ^ 16), xb = __shfl(a, lane ^ 16);
    a = odd ? xa : a;
    b = odd ? b : xb;
#endif
}
HV_DEV unsigned hv_pk_max_u16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(hv_u16x2, a), __builtin_bit_cast(hv_u16x2, b)));
}

struct HvAttn40Geom {
    static constexpr int D = 40, NW = 8, BQ = 32 * NW;
    static constexpr int KRS = 112;  // K row pitch: 80 data bytes + the augmented column (bf16 1.0 at byte 80) + padding; 28 dwords:
                                     // the 16 rows of a ds_read_b128 lane group (equal column offset in the 32x32 A-operand read)
                                     // fall on 16 different 4-dword bank windows (28 r mod 64 is a permutation of the multiples of 4)
    static constexpr int VRS = 160;  // V^T row pitch (64 keys = 128 data bytes): conflict-free for the 16x16x32 A-operand read
    static constexpr int DV = 48;    // V^T rows: 40 channels, the row of ones (denominator), 7 zero rows
    static constexpr int KBYTES = 64 * KRS, VBYTES = DV * VRS;
    static constexpr int KCH = 64 * 5, VCH = 40 * 8;  // 16-byte chunks of a K / V^T tile: 320 each
    static constexpr int VT0 = 512 - VCH;              // first thread of the V^T loaders (threads 192 .. 511; K: threads 0 .. 319)
    static constexpr int NBUF = 3;                     // tile buffers (K, V^T pairs): the two-group loop keeps a pair for four phases
};

// Per-thread context shared by the kernel and its careful fallback: indices, the query fragments, tile staging.
template <bool MASK>
struct HvAttn40Thread {
    using G = HvAttn40Geom;
    static constexpr int D = G::D;
    int tid, lane, wave, r16, quad, l32, half;
    int img, head, q_wave, sel, T1, ntiles;
    unsigned char *Ks, *Vs;
    bf16x8 qf[3];  // B operand of S^T = K.Q^T (32x32x16): lane = query l32, elements = head-dim 16 s + 8 half + 0..7; pre-multiplied
                   // by scale * log2(e) (scores come out of the MFMA in the exp2 domain); step 2 of the upper half is the augmented
                   // part: element 0 carries -m (the query's reference maximum, negated), the rest is zero
    float mneg;    // -m of this lane's query: always exactly representable in bf16 (it is what qf[2][0] of the upper half holds)
    // tile staging: one 16-byte chunk of the K tile (threads 0 .. 319: waves 0-4) and / or one of the V^T tile (threads
    // 192 .. 511: waves 3-7) per thread, through registers (the loads of a later tile fly during the arithmetic of this one)
    bool k_loader, v_loader;  // (wave-uniform)
    int krow, vcol, klds, vlds;
    u32x4 kreg, vreg;
    const char *kbase, *vbase;
    unsigned kstep, koff, voff, ldk_b, ldv_b;
    int src_kv0, src_L, vrow63, kcol;

    HV_DEV void setup(const hv_attention_params& p, unsigned char* smem, int qb, int head_, int img_) {
        tid = threadIdx.x, lane = tid & 63;
#ifndef HV_EMU
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#else
        wave = tid >> 6;
#endif
        r16 = lane & 15, quad = lane >> 4, l32 = lane & 31, half = lane >> 5;
        img = img_, head = head_;
        q_wave = qb * G::BQ + wave * 32;
        sel = (p.bank_sel != nullptr && p.L2 > 0) ? p.bank_sel[img] : -1;
        T1 = (p.L1 + 63) / 64;
        ntiles = T1 + (sel >= 0 ? (p.L2 + 63) / 64 : 0);
        Ks = smem;
        Vs = smem + G::NBUF * G::KBYTES;
        k_loader = wave < 5, v_loader = wave >= 3;
        krow = tid / 5;                                  // key row within the tile
        kcol = (tid - 5 * krow) * 16;                    // byte within its 80-byte head slice
        const int vid = tid - G::VT0, vrow = vid >> 3;   // V^T channel row
        vcol = (vid & 7) * 16;                           // byte within its 128 bytes (8 keys per chunk)
        vrow63 = vrow & 63;                              // (& 63: non-loader threads stay inside the 24-bit multiply)
        // key 16 b + 4 a + c of a 32-key block is staged at row 8 a + 4 b + c (see the header)
        klds = ((krow & 32) | (((krow >> 2) & 3) << 3) | (((krow >> 4) & 1) << 2) | (krow & 3)) * G::KRS + kcol;
        vlds = vrow * G::VRS + vcol;
        kreg = vreg = u32x4{0u, 0u, 0u, 0u};
        kbase = vbase = nullptr;
        kstep = koff = voff = 0;
        src_kv0 = src_L = 0;
        mneg = 0.f;
    }
    // LDS: zero everything once (the padding is never overwritten by the tile stores), then the augmented K column (bf16 1.0
    // at head-dim index 40 of every key row) and the V^T row of ones (row 40: the P.V MFMA accumulates the denominator)
    HV_DEV void init_lds() {
        for (int i = tid; i < G::NBUF * (G::KBYTES + G::VBYTES) / 16; i += 512) hv_st16(Ks + i * 16, u32x4{0u, 0u, 0u, 0u});
        __syncthreads();
        if (tid < 64 * G::NBUF) *reinterpret_cast<unsigned*>(Ks + (tid >> 6) * G::KBYTES + (tid & 63) * G::KRS + 2 * D) = 0x00003F80u;
        if (tid >= 256 && tid < 256 + 8 * G::NBUF)
            hv_st16(Vs + ((tid - 256) >> 3) * G::VBYTES + D * G::VRS + ((tid - 256) & 7) * 16,
                    u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});
        __syncthreads();
    }
    HV_DEV void load_q(const hv_attention_params& p) {
        const int q = q_wave + l32;
        const bf16_t* qrow = p.Q + ((long)img * p.Lq + q) * p.ldq + head * D;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (q < p.Lq && 16 * s + 8 * half + 8 <= D) v = hv_ld16(qrow + 16 * s + 8 * half);
            float f[8];
            hv_unpack8(v, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] *= p.scale * 1.44269504089f;
            qf[s] = hv_as_bf16x8(hv_pack8(f));
        }
        mneg = 0.f;
    }
    HV_DEV void set_reference(float mneg_new) {  // mneg_new: a bf16 value
        mneg = mneg_new;
        if (half) qf[2][0] = (short)hv_f2bf(mneg_new);
    }
    HV_DEV void set_source(const hv_attention_params& p, bool bank) {
        const unsigned rowbase = bank ? (unsigned)sel * (unsigned)p.L2 : (unsigned)img * (unsigned)p.L1;
        const unsigned ldk2 = (unsigned)(bank ? p.ldk2 : p.ldk) * 2u, ldv2 = (unsigned)(bank ? p.ldvt2 : p.ldvt) * 2u;  // bytes
        kbase = reinterpret_cast<const char*>(bank ? p.K2 : p.K) + ((size_t)rowbase * ldk2 + (size_t)(head * D) * 2u);
        vbase = reinterpret_cast<const char*>(bank ? p.Vt2 : p.Vt) + ((size_t)(head * D) * ldv2 + (size_t)rowbase * 2u);
        kstep = 64u * ldk2;
        src_kv0 = 0;
        src_L = bank ? p.L2 : p.L1;
        koff = hv_umul24((unsigned)krow, ldk2) + (unsigned)kcol;
        voff = hv_umul24((unsigned)vrow63, ldv2) + (unsigned)vcol;
    }
    // tiles are requested in order: 0, 1, 2, ...
    HV_DEV void load_tile(const hv_attention_params& p, int ti) {
        if (ti == 0) set_source(p, false);
        if (ti == T1) set_source(p, true);  // (rare, wave-uniform) the bank follows the own keys
        if (k_loader) {
            kreg = u32x4{0u, 0u, 0u, 0u};
            if (!MASK || src_kv0 + krow < src_L) kreg = hv_ld16(kbase + koff);
        }
        if (v_loader) {
            vreg = u32x4{0u, 0u, 0u, 0u};
            if (!MASK || src_kv0 + (vcol >> 1) < src_L) vreg = hv_ld16(vbase + voff);  // (L % 8 == 0: a chunk never straddles the end)
        }
        kbase += kstep;
        vbase += 128;
        src_kv0 += 64;
    }
    HV_DEV void store_tile(int buf) {
        if (k_loader) hv_st16(Ks + buf * G::KBYTES + klds, kreg);
        if (v_loader) hv_st16(Vs + buf * G::VBYTES + vlds, vreg);
    }
    // S^T of tile ti (two 32-key blocks x this wave's 32 queries) from K buffer `buf`, relative to the reference maximum
    HV_DEV void scores(const hv_attention_params& p, int ti, int buf, f32x16 (&sc)[2]) {
        const unsigned char* kb = Ks + buf * G::KBYTES + l32 * G::KRS + half * 16;
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) {
            f32x16 a;
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = 0.f;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const bf16x8 kf = hv_as_bf16x8(hv_ld16(kb + (32 * kbk) * G::KRS + s * 32));
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], a, 0, 0, 0);
            }
            sc[kbk] = a;
        }
        if (MASK) {
            const int kv0 = (ti >= T1 ? ti - T1 : ti) * 64, L = ti >= T1 ? p.L2 : p.L1;
            if (kv0 + 64 > L) {
#pragma unroll
                for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (kv0 + 32 * kbk + 16 * half + i >= L) sc[kbk][i] = -INFINITY;  // register i of half b = key 16 b + i
            }
        }
    }
    // packed probabilities w[j] = keys 16 half + 2 j, + 1 of a 32-key block -> its two B operands of the 16x16x32 MFMA
    // (query tiles 0-15 / 16-31): four lane-row swaps
    HV_DEV void to_operand(const unsigned (&w)[8], bf16x8& p0, bf16x8& p1) {
        u32x4 b0, b1;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            unsigned a = w[d], b = w[d + 4];
            hv_swap16_pair(a, b);
            b0[d] = a;
            b1[d] = b;
        }
        p0 = hv_as_bf16x8(b0);
        p1 = hv_as_bf16x8(b1);
    }
    // O^T += V^T . P^T from V^T buffer `buf` (row 40 of V^T is all ones: accumulates the denominator)
    HV_DEV void accumulate(int buf, const bf16x8 (&pf)[2][2], f32x4 (&oacc)[2][3]) {
        const unsigned char* vb = Vs + buf * G::VBYTES + r16 * G::VRS + quad * 16;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk) {
                const bf16x8 vf = hv_as_bf16x8(hv_ld16(vb + (16 * dt) * G::VRS + kbk * 64));
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
                    oacc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qt][kbk], oacc[qt][dt], 0, 0, 0);
            }
    }
    // normalise and store: lane owns query 16 qt + r16, channels 16 dt + 4 quad + 0..3; the denominator is O^T row 40
    // (fragment 2, rows 8 .. 11 of it = quad 2, register 0)
    HV_DEV void store_output(const hv_attention_params& p, const f32x4 (&oacc)[2][3]) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const float l = __shfl(oacc[qt][2][0], 32 + r16);
            const float inv = 1.0f / l;
            const int q = q_wave + 16 * qt + r16;
            if (q >= p.Lq) continue;
            bf16_t* dst = p.O + ((long)img * p.Lq + q) * p.ldo + head * D;
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                const int d = 16 * dt + 4 * quad;
                if (d < D) {
                    const u32x2 o = {hv_pack2(oacc[qt][dt][0] * inv, oacc[qt][dt][1] * inv),
                                     hv_pack2(oacc[qt][dt][2] * inv, oacc[qt][dt][3] * inv)};
                    hv_st8(dst + d, o);
                }
            }
        }
    }
};

HV_DEV void hv_attn40_prio(int v) {
#ifndef HV_EMU
    if (v) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
#endif
}

// The careful form, for a workgroup whose optimistic pass overflowed (rare: a later score exceeds the first tile's maximum by
// ~2^7 in the exp2 domain): single-phase loop, "some probability of the wave exceeds 2^THR" test per tile on the packed bf16
// pairs (positive bf16 order like 16-bit unsigned integers), reference raised and O^T rescaled once when it fires.  Starts the
// workgroup's work over and stores its output.  Not inlined: the kernel's registers are budgeted for the fast loop alone (the
// LDS pointer arrives as a generic pointer, so this path runs on flat loads -- it is not the one that is timed).
template <bool MASK>
__device__ __attribute__((noinline)) void hv_attention40_careful(const hv_attention_params& p, unsigned char* smem, int qb, int head,
                                                                 int img) {
    using G = HvAttn40Geom;
    HvAttn40Thread<MASK> c;
    c.setup(p, smem, qb, head, img);
    c.load_q(p);
    f32x4 oacc[2][3];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) oacc[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    c.load_tile(p, 0);
    for (int ti = 0; ti < c.ntiles; ++ti) {
        const int buf = ti & 1;
        c.store_tile(buf);
        __syncthreads();
        if (ti + 1 < c.ntiles) c.load_tile(p, ti + 1);
        f32x16 sacc[2];
        c.scores(p, ti, buf, sacc);
        unsigned w[2][8];
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                w[kbk][j] = hv_pack2(__builtin_amdgcn_exp2f(sacc[kbk][2 * j]), __builtin_amdgcn_exp2f(sacc[kbk][2 * j + 1]));
        unsigned pm = hv_pk_max_u16(w[0][0], w[0][1]);
#pragma unroll
        for (int j = 2; j < 8; ++j) pm = hv_pk_max_u16(pm, w[0][j]);
#pragma unroll
        for (int j = 0; j < 8; ++j) pm = hv_pk_max_u16(pm, w[1][j]);
        const bool over = __any((pm & 0xffffu) > 0x4380u || (pm >> 16) > 0x4380u);  // bf16 256.0 = 0x4380; +inf = 0x7F80 is above as well
        const bool first = ti == 0;  // the first tile fixes the reference maximum (it starts at 0, not at a score)
        if (first || over) {
            // raise the reference maximum by the query's tile maximum, redo the exponentials against it and scale what is
            // still at the old reference (O^T with its denominator row) exactly once
            float mx = sacc[0][0];
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[kbk][i]);
            mx = fmaxf(mx, hv_swap32(mx));  // the other half of the keys of the same query sits in lane ^ 32
            const float inc = first ? mx : fmaxf(mx, 0.f);
            const float mneg_new = hv_bf2f(hv_f2bf(c.mneg - inc));  // the new reference as the Q operand will carry it
            const float inc_eff = c.mneg - mneg_new;
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    w[kbk][j] = hv_pack2(__builtin_amdgcn_exp2f(sacc[kbk][2 * j] - inc_eff), __builtin_amdgcn_exp2f(sacc[kbk][2 * j + 1] - inc_eff));
            if (!first) {
                const float alpha = __builtin_amdgcn_exp2f(-inc_eff);
                const float a0 = __shfl(alpha, c.r16), a1 = __shfl(alpha, 16 + c.r16);  // lanes 0-31 hold queries 0-31 of the wave
#pragma unroll
                for (int dt = 0; dt < 3; ++dt) {
                    oacc[0][dt] *= a0;
                    oacc[1][dt] *= a1;
                }
            }
            c.set_reference(mneg_new);
        }
        bf16x8 pf[2][2];
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) c.to_operand(w[kbk], pf[0][kbk], pf[1][kbk]);
        c.accumulate(buf, pf, oacc);
    }
    c.store_output(p, oacc);
}

// Optional phase clock (tools/attn40_trace.hip): every wave of workgroup HV_ATTN40_TRACE sums the shader cycles it spends
// staging, in M phases, in V phases and at the barriers (scalar s_memtime differences: no memory traffic until the end).
#ifdef HV_ATTN40_TRACE
__device__ unsigned long long g_hv_a40_trace[8 * 8];
#define HV_A40_CLK(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define HV_A40_ADD(acc, a, b) acc += (b) - (a)
#else
#define HV_A40_CLK(var)
#define HV_A40_ADD(acc, a, b)
#endif

template <bool MASK>
__global__ __launch_bounds__(512, 4) void hv_attention40_kernel(hv_attention_params p) {
    using G = HvAttn40Geom;
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::NBUF * (G::KBYTES + G::VBYTES)];

    const int nqb = (p.Lq + G::BQ - 1) / G::BQ;
    const int total = nqb * p.heads * p.n_images;
    const int cpx = gridDim.x / 8;
    int t = (blockIdx.x % 8) * cpx + blockIdx.x / 8;  // XCD x walks the contiguous range [x cpx, (x + 1) cpx)
    if (t >= total) return;
    // the query blocks of an (image, head) run next to each other: its K / V^T stream through L2 once.  (The other raster --
    // the 8 heads of a query block adjacent, so that their 80-byte output pieces fill whole lines in L2 -- measured 2 % slower
    // on MI355X, profiles/r04_s1.txt: this kernel moves 0.45 TB/s, it is not bound by its HBM bytes.)
    const int qb = t % nqb;
    t /= nqb;
    const int head = t % p.heads;
    int img = t / p.heads;
    // alternate between the CFG halves (the conditional images attend to twice the keys): every XCD gets the same mix
    if ((p.n_images & 1) == 0) img = (img & 1) * (p.n_images >> 1) + (img >> 1);

    HvAttn40Thread<MASK> c;
    c.setup(p, smem, qb, head, img);
    c.init_lds();
    c.load_q(p);
    const int ntiles = c.ntiles;

    // ---- Two-group loop, optimistic reference.  A wave's work per 64-key tile is an MFMA phase (PV of the tile whose
    // probabilities it holds, then the scores of the next tile: 384 matrix-pipe cycles) and a VALU phase (32 exponentials at
    // ~8.5 cycles, 16 packs at ~4.7, 8 lane-row swaps at ~8.5: ~420 cycles; tools/valu_rate.hip, profiles/r04_s9_issue_rates.txt).
    // In a single-phase loop (barrier, S, P, PV per tile: the first round-4 version) the two waves of a workgroup that share
    // a SIMD -- wave w and w + 4 -- are in the same phase at the same time and take turns at the same unit; only the other
    // workgroup of the CU fills the idle one, when it happens to be out of phase: 778 cycles per wave-tile and SIMD were
    // measured, the SUM of the two units' times, although an MFMA stream and a v_exp stream of two waves on one SIMD
    // co-issue almost for free (same measurement).  Here waves 0-3 and waves 4-7 run half a tile apart -- one group's MFMA
    // phase beside the other's VALU phase on every SIMD, by construction -- with a workgroup barrier between phases:
    //   phase  -1      0      1      2      3     ...   2T-1    2T          M(t) = PV(t), then S(t + 1);  V(t) = P(t) from S(t)
    //   A:    M(-1)   V(0)   M(0)   V(1)   M(1)   ...  M(T-1)   -           M(-1) also fixes the reference maximum
    //   B:     -     M(-1)   V(0)   M(0)   V(1)   ...  V(T-1)  M(T-1)
    // Tile pair t (K, V^T) is read in phases 2t-1 .. 2t+2 (K by M(t-1), V^T by M(t)) and stored at the end of phase 2t-3
    // into buffer t % 3, whose previous pair t-3 was last read in phase 2t-4.
    const int grp = c.wave >> 2;
    f32x4 oacc[2][3];  // O^T accumulators [16-query tile][16-row channel fragment]: lane = query r16, channels 16 dt + 4 quad + 0..3
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) oacc[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x16 sacc[2];
    bf16x8 pf[2][2];
    int kbuf = 0, vbuf = 0, m_t = -1;  // M phase state: buffer of K(m_t + 1), of V^T(m_t)
    c.load_tile(p, 0);
    c.store_tile(0);
    if (ntiles > 1) c.load_tile(p, 1);
    int sbuf = 1;  // buffer of the pair staged next
    __syncthreads();
#ifdef HV_ATTN40_TRACE
    unsigned long long tr_stage = 0, tr_m = 0, tr_v = 0, tr_bar = 0, tr_nm = 0, tr_nv = 0;
    const unsigned long long tr_begin = __builtin_amdgcn_s_memtime();
#endif
    for (int ph = -1; ph <= 2 * ntiles; ++ph) {
        HV_A40_CLK(c1);
        const int i = ph + 1 - grp;  // this group's position in the sequence M(-1) V(0) M(0) ... V(T-1) M(T-1)  (wave-uniform)
        if (i >= 0 && i <= 2 * ntiles) {
            if (i & 1) {  // V phase: P from S (exp2 domain), into the B-operand layout of the PV MFMA
#pragma unroll
                for (int kbk = 0; kbk < 2; ++kbk) {
                    unsigned w[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        w[j] = hv_pack2(__builtin_amdgcn_exp2f(sacc[kbk][2 * j]), __builtin_amdgcn_exp2f(sacc[kbk][2 * j + 1]));
                    c.to_operand(w, pf[0][kbk], pf[1][kbk]);
                }
            } else {  // M phase
                hv_attn40_prio(1);
                if (m_t >= 0) c.accumulate(vbuf, pf, oacc);
                if (m_t + 1 < ntiles) c.scores(p, m_t + 1, kbuf, sacc);
                hv_attn40_prio(0);
                if (m_t < 0) {  // the first tile fixes the reference maximum: the query's tile maximum, as the Q operand will carry it
                    float mx = sacc[0][0];
#pragma unroll
                    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sacc[kbk][e]);
                    mx = fmaxf(mx, hv_swap32(mx));  // the other half of the keys of the same query sits in lane ^ 32
                    const float mneg_new = hv_bf2f(hv_f2bf(c.mneg - mx));
                    const float inc_eff = c.mneg - mneg_new;
#pragma unroll
                    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                        for (int e = 0; e < 16; ++e) sacc[kbk][e] -= inc_eff;
                    c.set_reference(mneg_new);
                }
                vbuf = kbuf;
                kbuf = kbuf == G::NBUF - 1 ? 0 : kbuf + 1;
                ++m_t;
            }
#ifdef HV_ATTN40_TRACE
            const unsigned long long c2 = __builtin_amdgcn_s_memtime();
            if (i & 1) tr_v += c2 - c1, ++tr_nv;
            else tr_m += c2 - c1, ++tr_nm;
#endif
        }
        HV_A40_CLK(c0);
        // Staging at the END of phase 2t - 3 (behind the phase's own instructions: an M wave's MFMAs are queued, the LDS stores
        // and the next requests issue beside them): pair t = (ph + 3) / 2 goes to its buffer, pair t + 1 is requested
        if (ph & 1) {
            const int ts = (ph + 3) >> 1;
            if (ts < ntiles) {
                c.store_tile(sbuf);
                sbuf = sbuf == G::NBUF - 1 ? 0 : sbuf + 1;
                if (ts + 1 < ntiles) c.load_tile(p, ts + 1);
            }
        }
        HV_A40_CLK(c3);
        HV_A40_ADD(tr_stage, c0, c3);
        __syncthreads();
        HV_A40_CLK(c4);
        HV_A40_ADD(tr_bar, c3, c4);
    }
#ifdef HV_ATTN40_TRACE
    if (blockIdx.x == HV_ATTN40_TRACE && c.lane == 0) {
        unsigned long long* o = g_hv_a40_trace + c.wave * 8;
        o[0] = __builtin_amdgcn_s_memtime() - tr_begin, o[1] = tr_stage, o[2] = tr_m, o[3] = tr_v, o[4] = tr_bar, o[5] = tr_nm, o[6] = tr_nv,
        o[7] = (unsigned long long)ntiles;
    }
#endif
    // ---- did anything overflow?  (inf / NaN anywhere in this lane's O^T columns, denominator row included: 0 x inf = NaN)
    float chk = 0.f;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) chk += (oacc[qt][dt][0] + oacc[qt][dt][1] + oacc[qt][dt][2] + oacc[qt][dt][3]) * 0.f;
    const int wave_bad = __any(chk != chk);
    // (every wave is behind the loop's last barrier: the first bytes of K buffer 0 -- tile data -- take the votes)
    if (c.lane == 0) reinterpret_cast<int*>(smem)[c.wave] = wave_bad;
    __syncthreads();
    int any_bad = 0;
#pragma unroll
    for (int w8 = 0; w8 < G::NW; ++w8) any_bad |= reinterpret_cast<const int*>(smem)[w8];
    if (any_bad) {  // (workgroup-uniform)
        __syncthreads();  // the votes are read before the careful pass stores its first tile over them
        const hv_attention_params pc = p;  // (a copy: taking the address of the kernel argument itself would move it to scratch for the fast path too)
        hv_attention40_careful<MASK>(pc, smem, qb, head, img);
        return;
    }
    c.store_output(p, oacc);
}

// tuning knob (hv_set_tuning): 1 = this kernel for head dim 40 (default), 0 = the generic kernel
static int g_hv_attn40 = 1;

static inline void hv_attention40_launch(const hv_attention_params& p, hipStream_t stream) {
    using G = HvAttn40Geom;
    const int total = ((p.Lq + G::BQ - 1) / G::BQ) * p.heads * p.n_images;
    const int grid = ((total + 7) / 8) * 8;
    const bool ragged = (p.L1 % 64) != 0 || (p.L2 % 64) != 0;
    hv_note("hv_attention40_kernel | n=%d heads=%d D=%d Lq=%d L1=%d L2=%d bank=%d", p.n_images, p.heads, p.D, p.Lq, p.L1, p.L2,
            p.bank_sel != nullptr && p.L2 > 0);
    if (ragged) hv_launch(hv_attention40_kernel<true>, dim3(grid), dim3(512), stream, p);
    else hv_launch(hv_attention40_kernel<false>, dim3(grid), dim3(512), stream, p);
}
