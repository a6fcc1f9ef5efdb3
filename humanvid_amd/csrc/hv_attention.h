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
//  * the head dim is not a multiple of 32 for d = 40 / 80: the remainder of the QK^T reduction is a zero-padded
//    32-deep step (d = 40 costs 64, d = 80 costs 96): every MFMA of an accumulation chain has the same shape -- a
//    16-deep MFMA chained to 32-deep ones reads a partially written accumulator on gfx950 with hipcc / ROCm 7.2
//    (profiles/r02_mfma_chain_hazard.md).
//  * key rows are staged into LDS in a permuted order (bits 2 and 3-4 rotated) so that the
//    probabilities of two adjacent score fragments concatenate, in-register, into the B operand of
//    O^T += V^T.P^T with natural key order -- no cross-lane traffic between the two matmuls.
//  * V arrives already transposed ([channel][token], written by the QKV GEMM epilogue), so both
//    K and V^T fragments are single LDS vector reads; LDS rows are padded to an odd number of
//    16-byte slots (conflict-free ds_read_b128).
//  * deferred-rescale softmax in the exp2 domain: the queries are pre-multiplied by scale * log2(e), the query's
//    reference maximum enters the QK^T MFMA as the accumulator's initial value (C = -m: the MFMA delivers s - m), and
//    the reference is only raised -- O^T and the denominator row scaled once -- in tiles where some probability of the
//    wave exceeds 2^THR (wave-uniform branch; forced in tests/kernel_cases.py); the ragged-tail masking is a separate
//    template instance.
//  * K/V tiles (64 keys) are double-buffered through registers, one barrier per tile; online
//    softmax in the exp2 domain; fp32 accumulation; register budget sized for >= 2 workgroups per CU.
#pragma once
#include "hv_common.h"
#include "humanvid_hip.h"
#include "hv_attention40.h"

#ifndef HV_ATTN_THR
#define HV_ATTN_THR 8.0f  // log2 units: probabilities stay below 2^THR before the reference maximum is raised
#endif
#define HV_ATTN_PTHR 256.0f  // = 2^HV_ATTN_THR, the same threshold on the probabilities
static_assert(HV_ATTN_THR == 8.0f, "HV_ATTN_PTHR must be 2^HV_ATTN_THR");

template <int D, int QT>
struct HvAttnGeom {
    static constexpr int NFULL = (D + 31) / 32;       // 32-deep QK^T steps (the last one zero-padded)
    static constexpr int DT = (D + 15) / 16;          // 16-row fragments of V^T / O^T
    static constexpr int DK = 32 * NFULL;
    static constexpr int DV = 16 * DT;
    static constexpr bool ONES = DV > D;  // spare V^T row available: a row of ones, so that the P.V MFMA that is issued anyway also
                                          // accumulates sum(P) (d = 40 pads V^T to 48 rows) and the 16 VALU adds per query fragment and tile disappear
    // LDS row strides (bytes): = 32 (mod 64).  A ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19,
    // 28-31}, ... (MI355X_MICROARCH.md, LDS table): a group mixes two quads of a fragment read, i.e. rows r16 at byte offset
    // 16 q and rows r16' at 16 (q + 1).  With the "odd multiple of 16 bytes" stride of rounds 1-2 (144) every fragment read
    // was a 2-way bank conflict under that grouping; strides of 96, 160, 224, ... are conflict-free (checked per group with
    // the documented bank rule, the same model that reproduces the conflict-free GEMM swizzle).
    // (d = 80 keeps 208 / 144: with 224 / 160 only two workgroups fit a CU instead of three, measured +17 %)
    static constexpr bool S32 = D != 80;
    static constexpr int KRS = S32 ? ((DK * 2 + 31) / 64) * 64 + 32 : DK * 2 + 16;  // K rows: DK * 2 data bytes
    static constexpr int VRS = S32 ? 160 : 64 * 2 + 16;                                // V^T rows: 64 keys
    static constexpr int KBYTES = 64 * KRS;
    static constexpr int VBYTES = DV * VRS;
    static constexpr int KCH = 64 * (D / 8);          // 16-byte chunks of a K tile
    static constexpr int VCH = D * 8;                 // 16-byte chunks of a V^T tile
    // a tile = KCH (= VCH) chunks over 256 threads: FIT whole 16-byte chunks per thread, and the REM remaining chunks cut
    // into 256 equal pieces of PB bytes (d = 40: 64 chunks -> 4 bytes per thread, d = 80: 128 -> 8, d = 160: none) -- every
    // thread issues the same loads / LDS stores (round 2 gave the remainder to wave 0 as whole chunks: a second, mostly
    // idle 16-byte register set and address pair per operand in every thread, and exec-masked branches in every tile)
    static constexpr int FIT = KCH / 256;
    static constexpr int REM = KCH % 256;
    static constexpr int PB = REM * 16 / 256;         // bytes of the remainder piece per thread (0, 4 or 8)
    static constexpr int PPC = PB ? 16 / PB : 1;      // pieces per chunk
    static_assert(KCH == VCH && (REM == 0 || REM == 64 || REM == 128), "tile split");
    static constexpr int BQ = 4 * 16 * QT;            // queries per workgroup
};

// waves per SIMD the register allocator is asked to fit (LDS allows 5 / 3 / 1 workgroups per CU for
// d = 40 / 80 / 160): more resident waves let one wave's softmax VALU overlap another's MFMA
template <int D, int QT>
struct HvAttnOcc {
    static constexpr int value = (D == 40 && QT == 2) ? 4 : ((D == 80 && QT == 2) ? 2 : 1);
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
    const int q_wave = qb * G::BQ + wave * 16 * QT;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int q = q_wave + 16 * qt + r16;
        const bf16_t* qrow = p.Q + ((long)img * p.Lq + q) * p.ldq + head * D;
#pragma unroll
        for (int s = 0; s < NFULL; ++s) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (q < p.Lq && 32 * s + 8 * quad + 8 <= D) v = hv_ld16(qrow + 32 * s + 8 * quad);
            {  // scores come out of the MFMA in the exp2 domain
                float f[8];
                hv_unpack8(v, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] *= p.scale * 1.44269504089f;
                v = hv_pack8(f);
            }
            qf[qt][s] = hv_as_bf16x8(v);
        }
    }

    u32x4 kreg[G::FIT > 0 ? G::FIT : 1], vreg[G::FIT > 0 ? G::FIT : 1];
    u32x2 kpc = {0u, 0u}, vpc = {0u, 0u};  // remainder pieces (PB bytes each)
    // Per-lane parts of the K / V^T tile addresses, computed once (32-bit byte offsets from the wave-uniform tensor base:
    // hv_attention_launch checks the spans); per tile only a scalar offset changes.  The offsets for the bank tensors
    // (other row strides) are derived from (row, byte-in-row) when a bank tile is loaded: one v_mad per load.
    constexpr int CPRK = D / 8;  // 16-byte chunks per K row
    int krow[G::FIT + 1], kcol[G::FIT + 1], vrow[G::FIT + 1], vcol[G::FIT + 1];  // [FIT] = the remainder piece
#pragma unroll
    for (int i = 0; i <= G::FIT; ++i) {
        const int id = i < G::FIT ? tid + 256 * i : 256 * G::FIT + tid / G::PPC;
        const int sub = i < G::FIT ? 0 : (tid % G::PPC) * G::PB;
        krow[i] = id / CPRK, kcol[i] = (id % CPRK) * 16 + sub;
        vrow[i] = id >> 3, vcol[i] = (id & 7) * 16 + sub;
    }
    // Tile source state, advanced one 64-key tile per load: wave-uniform base pointers of the next K / V^T tile (SGPR pairs:
    // the loads take them as scalar bases, the per-lane part is a 32-bit offset register) and the per-lane offsets of the
    // CURRENT source tensor (own keys, then the bank: other row strides -> recomputed once at the switch).  Round 2
    // re-derived all of this from the kernel arguments in every tile: ~40 scalar instructions and four s_load round trips
    // (whose lgkmcnt wait also waits for the wave's LDS traffic) per tile.
    const char* kbase = nullptr;
    const char* vbase = nullptr;
    unsigned kstep = 0;  // bytes from one K tile to the next (64 rows)
    int src_kv0 = 0, src_L = 0;
    unsigned koff[G::FIT + 1], voff[G::FIT + 1];
    auto set_source = [&](bool bank) {
        const unsigned rowbase = bank ? (unsigned)sel * (unsigned)p.L2 : (unsigned)img * (unsigned)p.L1;
        const unsigned ldk2 = (unsigned)(bank ? p.ldk2 : p.ldk) * 2u, ldv2 = (unsigned)(bank ? p.ldvt2 : p.ldvt) * 2u;  // bytes
        kbase = reinterpret_cast<const char*>(bank ? p.K2 : p.K) + ((size_t)rowbase * ldk2 + (size_t)(head * D) * 2u);
        vbase = reinterpret_cast<const char*>(bank ? p.Vt2 : p.Vt) + ((size_t)(head * D) * ldv2 + (size_t)rowbase * 2u);
        kstep = 64u * ldk2;
        src_kv0 = 0;
        src_L = bank ? p.L2 : p.L1;
#pragma unroll
        for (int i = 0; i <= G::FIT; ++i) {
            koff[i] = hv_umul24(krow[i], ldk2) + (unsigned)kcol[i];
            voff[i] = hv_umul24(vrow[i], ldv2) + (unsigned)vcol[i];
        }
    };
    auto load_tile_inc = [&](int ti) {
        if (ti == T1) set_source(true);  // (rare, wave-uniform) the bank follows the own keys
        const int kv0 = src_kv0;
        const int L = src_L;
#pragma unroll
        for (int i = 0; i < G::FIT; ++i) {
            u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
            if (!MASK || kv0 + krow[i] < L) kv = hv_ld16(kbase + koff[i]);
            if (!MASK || kv0 + (vcol[i] >> 1) < L) vv = hv_ld16(vbase + voff[i]);
            kreg[i] = kv, vreg[i] = vv;
        }
        if (G::PB) {
            kpc = vpc = u32x2{0u, 0u};
            if (!MASK || kv0 + krow[G::FIT] < L) {
                if (G::PB == 8) kpc = hv_ld8(kbase + koff[G::FIT]);
                else kpc[0] = *reinterpret_cast<const unsigned*>(kbase + koff[G::FIT]);
            }
            if (!MASK || kv0 + (vcol[G::FIT] >> 1) < L) {  // (pieces never straddle the 8-key granule of the MASK contract)
                if (G::PB == 8) vpc = hv_ld8(vbase + voff[G::FIT]);
                else vpc[0] = *reinterpret_cast<const unsigned*>(vbase + voff[G::FIT]);
            }
        }
        kbase += kstep;
        vbase += 128;
        src_kv0 += 64;
    };
    auto load_tile_abs = [&](int ti) {  // d = 80 / 160: addresses from the tile index (measured faster there than the running pointers)
        const bool bank = ti >= T1;
        const int kv0 = (bank ? ti - T1 : ti) * 64;
        const int L = bank ? p.L2 : p.L1;
        const unsigned rowbase = bank ? (unsigned)sel * (unsigned)p.L2 : (unsigned)img * (unsigned)p.L1;
        const char* Kp = reinterpret_cast<const char*>(bank ? p.K2 : p.K);
        const char* Vp = reinterpret_cast<const char*>(bank ? p.Vt2 : p.Vt);
        // wave-uniform parts
        const unsigned ldk2 = (unsigned)(bank ? p.ldk2 : p.ldk) * 2u, ldv2 = (unsigned)(bank ? p.ldvt2 : p.ldvt) * 2u;  // bytes
        const unsigned kb = (rowbase + (unsigned)kv0) * ldk2 + (unsigned)(head * D) * 2u;
        const unsigned vb = (unsigned)(head * D) * ldv2 + (rowbase + (unsigned)kv0) * 2u;
#pragma unroll
        for (int i = 0; i < G::FIT; ++i) {
            u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
            if (!MASK || kv0 + krow[i] < L) kv = hv_ld16(Kp + (kb + hv_umul24(krow[i], ldk2) + (unsigned)kcol[i]));
            if (!MASK || kv0 + (vcol[i] >> 1) < L) vv = hv_ld16(Vp + (vb + hv_umul24(vrow[i], ldv2) + (unsigned)vcol[i]));
            kreg[i] = kv, vreg[i] = vv;
        }
        if (G::PB) {
            const char* ka = Kp + (kb + hv_umul24(krow[G::FIT], ldk2) + (unsigned)kcol[G::FIT]);
            const char* va = Vp + (vb + hv_umul24(vrow[G::FIT], ldv2) + (unsigned)vcol[G::FIT]);
            kpc = vpc = u32x2{0u, 0u};
            if (!MASK || kv0 + krow[G::FIT] < L) {
                if (G::PB == 8) kpc = hv_ld8(ka);
                else kpc[0] = *reinterpret_cast<const unsigned*>(ka);
            }
            if (!MASK || kv0 + (vcol[G::FIT] >> 1) < L) {  // (pieces never straddle the 8-key granule of the MASK contract)
                if (G::PB == 8) vpc = hv_ld8(va);
                else vpc[0] = *reinterpret_cast<const unsigned*>(va);
            }
        }
    };
    constexpr bool INC = D == 40;  // running-pointer loads: -2.8 % at d = 40, +12 % at d = 80 (same-box A/B, round 3)
    auto load_tile = [&](int ti) {
        if constexpr (INC) load_tile_inc(ti);
        else load_tile_abs(ti);
    };
    // key kv = 32a + 8b + 4c' + e  ->  LDS row 32a + 16c' + 4b + e
    auto key_row = [](int r) { return (r & ~0x1c) | ((r & 4) << 2) | ((r & 0x18) >> 1); };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < G::FIT; ++i) {
            hv_st16(Ks + buf * G::KBYTES + key_row(krow[i]) * G::KRS + kcol[i], kreg[i]);
            hv_st16(Vs + buf * G::VBYTES + vrow[i] * G::VRS + vcol[i], vreg[i]);
        }
        if (G::PB) {
            unsigned char* kd = Ks + buf * G::KBYTES + key_row(krow[G::FIT]) * G::KRS + kcol[G::FIT];
            unsigned char* vd = Vs + buf * G::VBYTES + vrow[G::FIT] * G::VRS + vcol[G::FIT];
            if (G::PB == 8) {
                hv_st8(kd, kpc);
                hv_st8(vd, vpc);
            } else {
                *reinterpret_cast<unsigned*>(kd) = kpc[0];
                *reinterpret_cast<unsigned*>(vd) = vpc[0];
            }
        }
    };

    f32x4 oacc[QT][DT];
    float lrun[QT];
    float mneg[QT];  // the query's reference maximum, negated: the QK^T accumulator's initial value; starts at 0
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        mneg[qt] = 0.f;
        lrun[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) oacc[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    if (INC) set_source(false);
    load_tile(0);
    __syncthreads();  // LDS initialisation complete before the first tile store
    for (int ti = 0; ti < ntiles; ++ti) {
        const int buf = ti & 1;
        store_tile(buf);
        __syncthreads();
        if (ti + 1 < ntiles) load_tile(ti + 1);
        const unsigned char* kb = Ks + buf * G::KBYTES + r16 * G::KRS;
        const unsigned char* vb = Vs + buf * G::VBYTES + r16 * G::VRS + quad * 16;

        // ---- S^T fragments: sacc[kvf][qt], lane = (query r16, quad), reg r <-> key
        //      kv = 32*(kvf>>1) + 8*quad + 4*(kvf&1) + r
        f32x4 sacc[QT][4];
#ifndef HV_EMU
        if (D <= 80) __builtin_amdgcn_s_setprio(1);  // MFMA clusters at raised priority: -1.7 % at d = 40 (four waves per SIMD)
#endif
#pragma unroll
        for (int kvf = 0; kvf < 4; ++kvf) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) sacc[qt][kvf] = f32x4{mneg[qt], mneg[qt], mneg[qt], mneg[qt]};  // lane = one query
#pragma unroll
            for (int s = 0; s < NFULL; ++s) {
                // (issuing all fragment reads of the tile ahead of their MFMAs -- 24-32 more registers, three waves per SIMD -- was
                //  measured 3 % slower at d = 40: the waits it removes are worth less than the fourth wave)
                const bf16x8 kf = hv_as_bf16x8(hv_ld16(kb + (16 * kvf) * G::KRS + s * 64 + quad * 16));
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    sacc[qt][kvf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][s], sacc[qt][kvf], 0, 0, 0);
            }
        }
        // ---- mask the ragged tail of this source (MASK instances only)
        auto mask_tail = [&](f32x4 (&sc)[4]) __attribute__((always_inline)) {
            if (!MASK) return;
            const bool bank = ti >= T1;
            const int kv0 = (bank ? ti - T1 : ti) * 64;
            const int L = bank ? p.L2 : p.L1;
            if (kv0 + 64 > L) {
#pragma unroll
                for (int kvf = 0; kvf < 4; ++kvf)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kv0 + 32 * (kvf >> 1) + 8 * quad + 4 * (kvf & 1) + r >= L) sc[kvf][r] = -INFINITY;
            }
        };
#ifndef HV_EMU
        if (D <= 80) __builtin_amdgcn_s_setprio(0);
#endif
        // ---- online softmax (exp2 domain) and P^T fragments
        bf16x8 pf[QT][2];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            // Exponentials against the current reference maximum first (the QK^T MFMA delivered s - m).  The tile only needs
            // the rescale branch when some probability exceeds 2^THR -- exp2 is monotonic, so the test runs on the
            // probabilities: their lane-local maximum is eight v_max3_f32 (exp2 results are canonical; fmaxf on the raw
            // MFMA outputs costs a canonicalising v_max x, x per operand in IEEE mode: 28 instructions for 16 values, and
            // an inline-asm v_max3 on MFMA outputs is not covered by hipcc's MFMA -> VALU hazard padding), and "some
            // query of this wave exceeds the threshold" is the same predicate over lane-local maxima as over the
            // cross-quad reduced ones -- the two ds_bpermute round trips of the reduction move into the rare branch.
            mask_tail(sacc[qt]);
            float pv[4][4];  // takes the place of the scores (they are recomputed in the rare branch: registers)
#pragma unroll
            for (int kvf = 0; kvf < 4; ++kvf)
#pragma unroll
                for (int r = 0; r < 4; ++r) pv[kvf][r] = __builtin_amdgcn_exp2f(sacc[qt][kvf][r]);
            float pm = fmaxf(fmaxf(pv[0][0], pv[0][1]), pv[0][2]);
            pm = fmaxf(fmaxf(pm, pv[0][3]), pv[1][0]);
            pm = fmaxf(fmaxf(pm, pv[1][1]), pv[1][2]);
            pm = fmaxf(fmaxf(pm, pv[1][3]), pv[2][0]);
            pm = fmaxf(fmaxf(pm, pv[2][1]), pv[2][2]);
            pm = fmaxf(fmaxf(pm, pv[2][3]), pv[3][0]);
            pm = fmaxf(fmaxf(pm, pv[3][1]), pv[3][2]);
            pm = fmaxf(pm, pv[3][3]);
            const bool first = ti == 0;  // the first tile fixes the reference maximum (it starts at 0, not at a score)
            if (first || __any(pm > HV_ATTN_PTHR)) {
                // rare: raise the reference maximum by the query's tile maximum (reduced over the four quads), redo the
                // exponentials against it and scale everything that is still at the old reference exactly once.  The
                // scores are recomputed from the K tile in LDS (keeping them live beside the probabilities on the common
                // path costs 16 registers per query fragment -- spills at four waves per SIMD).
#ifndef HV_EMU
                asm volatile("" ::: "memory");  // keeps the LDS reads (and with them the MFMAs) below from being hoisted out of the branch
#endif
                f32x4 s2[4];
#pragma unroll
                for (int kvf = 0; kvf < 4; ++kvf) {
                    s2[kvf] = f32x4{mneg[qt], mneg[qt], mneg[qt], mneg[qt]};
#pragma unroll
                    for (int s = 0; s < NFULL; ++s) {
                        const bf16x8 kf = hv_as_bf16x8(hv_ld16(kb + (16 * kvf) * G::KRS + s * 64 + quad * 16));
                        s2[kvf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][s], s2[kvf], 0, 0, 0);
                    }
                }
                mask_tail(s2);
                float mx = fmaxf(fmaxf(s2[0][0], s2[0][1]), fmaxf(s2[0][2], s2[0][3]));
#pragma unroll
                for (int kvf = 1; kvf < 4; ++kvf)
                    mx = fmaxf(fmaxf(fmaxf(mx, s2[kvf][0]), fmaxf(s2[kvf][1], s2[kvf][2])), s2[kvf][3]);
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float inc = first ? mx : fmaxf(mx, 0.f);
#pragma unroll
                for (int kvf = 0; kvf < 4; ++kvf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[kvf][r] = __builtin_amdgcn_exp2f(s2[kvf][r] - inc);
                if (!first) {
                    const float alpha = __builtin_amdgcn_exp2f(-inc);
                    if (!G::ONES) lrun[qt] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) oacc[qt][dt] *= alpha;
                }
                mneg[qt] -= inc;
            }
            if (!G::ONES) {
                float psum = 0.f;
#pragma unroll
                for (int kvf = 0; kvf < 4; ++kvf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) psum += pv[kvf][r];
                lrun[qt] += psum;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 w = {hv_pack2(pv[2 * ks][0], pv[2 * ks][1]), hv_pack2(pv[2 * ks][2], pv[2 * ks][3]),
                           hv_pack2(pv[2 * ks + 1][0], pv[2 * ks + 1][1]),
                           hv_pack2(pv[2 * ks + 1][2], pv[2 * ks + 1][3])};
                pf[qt][ks] = hv_as_bf16x8(w);
            }
        }
        // ---- O^T += V^T . P^T   (row D of V^T is all ones when ONES: accumulates the denominator)
#ifndef HV_EMU
        if (D <= 80) __builtin_amdgcn_s_setprio(1);  // MFMA clusters at raised priority: -1.7 % at d = 40 (four waves per SIMD)
#endif
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 vf = hv_as_bf16x8(hv_ld16(vb + (16 * dt) * G::VRS + ks * 64));
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    oacc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qt][ks], oacc[qt][dt], 0, 0, 0);
            }
#ifndef HV_EMU
        if (D <= 80) __builtin_amdgcn_s_setprio(0);
#endif
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

static inline int hv_attention_launch(const hv_attention_params& p, hipStream_t stream) {
    if (p.L1 <= 0 || p.L1 % 8 != 0 || p.L2 % 8 != 0 || p.Lq <= 0) return -1;
    if (p.ldq % 8 || p.ldk % 8 || p.ldvt % 8 || p.ldo % 4) return -1;
    if (p.L2 > 0 && p.bank_sel != nullptr && (!p.K2 || !p.Vt2 || p.ldk2 % 8 || p.ldvt2 % 8)) return -1;
    {  // row strides in bytes are 24-bit multiplier operands inside the kernel (hv_umul24)
        const long lim24 = 1L << 24;
        if (p.ldk * 2 >= lim24 || p.ldvt * 2 >= lim24) return -1;
        if (p.L2 > 0 && p.bank_sel != nullptr && (p.ldk2 * 2 >= lim24 || p.ldvt2 * 2 >= lim24)) return -1;
    }
    {  // 32-bit byte offsets inside the kernel
        const long lim = 1L << 32, C = (long)p.heads * p.D;
        if ((long)p.n_images * p.L1 * p.ldk * 2 >= lim || (C * p.ldvt + (long)p.n_images * p.L1) * 2 >= lim) return -1;
        if (p.L2 > 0 && p.bank_sel != nullptr && (64L * p.L2 * p.ldk2 * 2 >= lim || (C * p.ldvt2 + 64L * p.L2) * 2 >= lim))
            return -1;
    }
    switch (p.D) {
        case 40:
            if (g_hv_attn40) hv_attention40_launch(p, stream);  // the dedicated level-0 kernel (hv_attention40.h)
            else hv_attention_launch_t<40, 2>(p, stream);
            break;
        case 80: hv_attention_launch_t<80, 2>(p, stream); break;
        case 160: hv_attention_launch_t<160, 2>(p, stream); break;  // (one query fragment per wave measured slower in round 2: removed)
        default: return -2;
    }
    return 0;
}
