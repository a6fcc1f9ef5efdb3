// hv_attention2.h -- spatial self-attention with reference-bank keys, round-2 kernel (gfx950).
//
// Same semantics as hv_attention.h (diffusers SDPA as called by the read-mode patched transformer block,
// /root/reference/src/models/mutual_self_attention.py:147-186), different data path.  What the round-1 kernel lost its
// time on (profiles/README.md, attn_trace): per 64-key tile 1365 of 4700 cycles ISSUING global loads (a head's K rows are
// 80-byte pieces of 640-byte token rows), 611 cycles storing the tile to LDS from registers, and V needed a transposed
// copy that the QKV GEMM wrote with 2-byte scattered stores.  Here:
//
//  * one workgroup = 8 waves = a 160-channel slice (4 / 2 / 1 heads for d = 40 / 80 / 160) x 128 / 128 / 256 queries, one
//    wave per (head, query group), so K and V are consumed as 320-byte row pieces straight out of the row-major
//    [token][q | k | v] tensor the QKV GEMM writes: no head-major re-layout, no V^T tensor.  (First version: all 8 heads x
//    64 queries, 640-byte pieces -- attn2_trace showed ~1000 cycles per tile just moving 44 KB L2 -> LDS through the
//    CU's load path; this shape moves 21 KB per tile for twice the queries);
//  * K / V tiles (32 keys) go global -> LDS by LDS-DMA (global_load_lds, 1 KiB per wave-instruction) into 3-slot
//    rings, two groups in flight across ONE barrier per tile with counted vmcnt; software pipeline: S^T of tile t + 1
//    (MFMA) is issued ahead of the softmax of tile t (VALU), so K runs one tile ahead of V; the LDS image is built by the per-lane
//    SOURCE addresses: K rows 21 chunks (336 B) apart -- the extra position is a dummy load --, V rows 320 B apart,
//    which makes both fragment reads bank-conflict free;
//  * S^T = K.Q^T and O^T += V^T.P^T on mfma_f32_32x32x16_bf16: a lane owns ONE query (column lane & 31) and 16 of the 32
//    keys of a tile, its partner lane (lane ^ 32) the other 16: row max = 15 in-register max + one permlane32_swap;
//    every accumulation chain uses a single MFMA shape (rule from the round-1 hazard, see hv_temporal.h); d = 40 pads the
//    reduction to 48 with zeroed query lanes;
//  * V^T fragments come from the row-major V tile through ds_read_b64_tr_b16 (transposing LDS read);
//  * K rows are read in the order kappa(i) = i with bits 2 and 3 swapped, so the probabilities a lane holds after S^T are
//    exactly the keys the B operand of V^T.P^T wants from it: no cross-lane traffic between the two matmuls;
//  * images are visited alternating between the two CFG halves, so every XCD gets the same mix of 1x (own keys) and 2x
//    (own + bank keys) work.
#pragma once
#include "hv_common.h"
#include "humanvid_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef HV_A2_DBG
#define HV_A2_DBG 0  // timing experiments (wrong results!): 1 no DMA inside the tile loop, 2 no exp / sum / pack, 4 no P.V MFMAs,
#endif               // 8 no S^T MFMAs inside the loop, 16 no per-tile barrier
// phase timestamps for tools/attn2_trace.hip (which includes hv_gemm.h first with HV_GEMM_TRACE defined)
#ifndef HV_TRACE
#define HV_TRACE(id)
#endif

template <int D>
struct HvAttn2Geom {
    static constexpr int CH = 160;                 // channels per workgroup (a 320-byte piece of every K / V row)
    static constexpr int HG = CH / D;              // heads per workgroup: 4 / 2 / 1
    static constexpr int QG = 8 / HG;              // query groups (waves per head): 2 / 4 / 8
    static constexpr int NQB = D == 40 ? 2 : 1;    // 32-query blocks per wave
    static constexpr int BQ = 32 * NQB * QG;       // queries per workgroup: 128 / 128 / 256
    static constexpr int TK = 32;                  // keys per tile
    static constexpr int KS = (D + 15) / 16;       // 16-deep steps of the QK^T reduction: 3 / 5 / 10
    static constexpr int MT = (D + 31) / 32;       // 32-row tiles of O^T: 2 / 3 / 5
    static constexpr int RCH = CH * 2 / 16;        // real 16-byte chunks per row piece (20)
    static constexpr int KCH = RCH + 1, VCH = RCH; // chunk positions per LDS row: K rows 336 B apart (odd multiple of 16:
                                                   // conflict-free ds_read_b128 by row), V rows 320 B (= 64 mod 256: the four
                                                   // rows of a transposing read land on disjoint bank quarters)
    static constexpr int KROW = KCH * 16, VROW = VCH * 16;
    static constexpr int KINST = (TK * KCH + 63) / 64, VINST = (TK * VCH + 63) / 64;  // wave-instructions per tile: 11 + 10
    static constexpr int NINST = KINST + VINST;
    static constexpr int KBYTES = KINST * 1024, STAGE = NINST * 1024;
    static constexpr int NST = 3;
    static constexpr int TAILPAD = 256;            // V^T fragment rows past D read up to 46 bytes beyond the last stage
    static constexpr int MAXI = (NINST + 7) / 8;   // DMA instructions per wave and tile (3)
};

template <int D>
__global__ __launch_bounds__(512, 2) void hv_attention2_kernel(hv_attention_params p) {
    using G = HvAttn2Geom<D>;
    constexpr int HG = G::HG, NQB = G::NQB, KS = G::KS, MT = G::MT, TK = G::TK;
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::NST * G::STAGE + G::TAILPAD];

    const int tid = threadIdx.x, lane = tid & 63;
#ifndef HV_EMU
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: scalar branches, SGPR LDS bases
#else
    const int wave = tid >> 6;
#endif
    const int col = lane & 31, half = lane >> 5;
    const int hw = wave % HG, qg = wave / HG;
#ifdef HV_GEMM_TRACE
    int hv_ti = 0;
#endif

    // ---- which (image, head group, query block): XCD-contiguous ranges, images alternating between the CFG halves
    const int nqb = (p.Lq + G::BQ - 1) / G::BQ;
    const int groups = p.heads / HG;
    const int total = nqb * groups * p.n_images;
    const int cpx = gridDim.x / 8;
    int t = (blockIdx.x % 8) * cpx + blockIdx.x / 8;
    if (t >= total) return;
    const int qb = t % nqb;
    t /= nqb;
    const int hgi = t % groups;
    int img = t / groups;
    if ((p.n_images & 1) == 0) img = (img & 1) * (p.n_images >> 1) + (img >> 1);
    const int sel = (p.bank_sel != nullptr && p.L2 > 0) ? p.bank_sel[img] : -1;
    const int T1 = (p.L1 + TK - 1) / TK;
    const int T2 = sel >= 0 ? (p.L2 + TK - 1) / TK : 0;
    const int ntiles = T1 + T2;
    const int c0 = hgi * G::CH;                    // first channel of this head group
    const int head = hgi * HG + hw;

    // ---- DMA plan of this wave: instruction i = wave + 8 n fills LDS chunks [64 i, 64 i + 64) of a stage.
    // group g of the DMA stream = { K tile g -> K slot g % 3 , V tile g - 1 -> V slot (g + 2) % 3 } (tile indices clamped
    // into [0, ntiles): the few redundant loads at both ends keep every group the same size, so the vmcnt bookkeeping is a
    // compile-time constant).  K runs one tile ahead of V because S^T of tile t + 1 is computed under the softmax of tile t.
    // Groups are issued strictly in order; each lane keeps its source POINTERS and advances them by one tile per group (two
    // VALU adds per wave-instruction) -- recomputing row * stride per tile cost ~170 cycles per DMA instruction (attn2_trace).
    const char* dma_src[G::MAXI];
#pragma unroll
    for (int n = 0; n < G::MAXI; ++n) dma_src[n] = nullptr;
    auto tile_full = [&](int ti) { return ti < T1 ? (ti + 1) * TK <= p.L1 : (ti - T1 + 1) * TK <= p.L2; };
    // source pointers of tile `ti` for the K (isk) or V instructions of this wave: absolute (slow) form
    auto seek = [&](int ti, bool isk) __attribute__((always_inline)) {
        const bool bank = ti >= T1;
        const int kv0 = (bank ? ti - T1 : ti) * TK;
        const int L = bank ? p.L2 : p.L1;
        const long rowbase = bank ? (long)sel * p.L2 : (long)img * p.L1;
        const char* base = reinterpret_cast<const char*>(isk ? (bank ? p.K2 : p.K) : (bank ? p.Vt2 : p.Vt)) + c0 * 2;
        const long ld = (isk ? (bank ? p.ldk2 : p.ldk) : (bank ? p.ldvt2 : p.ldvt)) * 2;
#pragma unroll
        for (int n = 0; n < G::MAXI; ++n) {
            const int i = wave + 8 * n;
            if ((i < G::KINST) == isk && i < G::NINST) {
                // LDS chunk j of the K / V region: key row j / per inside the tile, 16-byte piece j % per of the 640-byte row
                // piece (the pad position and rows past the tile are dummies: they re-load valid data)
                const int j = (isk ? i : i - G::KINST) * 64 + lane;
                const int per = isk ? G::KCH : G::VCH;
                const int row = min(j / per, TK - 1), colb = min(j % per, G::RCH - 1) * 16;
                dma_src[n] = base + (rowbase + min(kv0 + row, L - 1)) * ld + colb;
            }
        }
    };
    auto advance = [&](int ti, bool isk) __attribute__((always_inline)) {  // tile ti - 1 -> ti
        // straight-line fast path (one 64-bit add per pointer); the rare cases -- first tile of a source, ragged last tile --
        // overwrite the pointers afterwards, so no value has to be merged across a branch
        const bool bank = ti >= T1;
        const long stride = (isk ? (bank ? p.ldk2 : p.ldk) : (bank ? p.ldvt2 : p.ldvt)) * (2 * TK);
#pragma unroll
        for (int n = 0; n < G::MAXI; ++n) {
            const int i = wave + 8 * n;
            if ((i < G::KINST) == isk && i < G::NINST) dma_src[n] += stride;
        }
        if (ti == 0 || ti == T1 || !tile_full(ti)) seek(ti, isk);
    };
    auto issue_prepare = [&](int g) __attribute__((always_inline)) {  // move the source pointers to group g
        if (g == 0) {
            seek(0, true);
            seek(0, false);
        } else {
            if (g <= ntiles - 1) advance(g, true);                 // K tile g (clamped past the end: pointers stay)
            if (g >= 2 && g - 1 <= ntiles - 1) advance(g - 1, false);  // V tile g - 1 (group 1 re-sends V tile 0)
        }
    };
    auto issue_part = [&](int g, int n) __attribute__((always_inline)) {  // this wave's n-th instruction of group g
        const int i = wave + 8 * n;
        if (i < G::NINST) {
            unsigned char* st = smem + ((i < G::KINST ? g : g + 2) % G::NST) * G::STAGE;
            hv_glds16(dma_src[n], st + i * 1024);
        }
    };
    auto issue = [&](int g) __attribute__((always_inline)) {
        issue_prepare(g);
#pragma unroll
        for (int n = 0; n < G::MAXI; ++n) issue_part(g, n);
    };
    const int my_inst = (G::NINST - wave + 7) / 8;  // 6 for waves 0..2, 5 otherwise

    // ---- query fragments (B operand of S^T = K.Q^T): lane = query `col`, reduction slice 8 half .. 8 half + 7
    bf16x8 qf[NQB][KS];
    const int q_wave = qb * G::BQ + qg * 32 * NQB;
#pragma unroll
    for (int b = 0; b < NQB; ++b) {
        const int q = min(q_wave + 32 * b + col, p.Lq - 1);
        const bf16_t* qrow = p.Q + ((long)img * p.Lq + q) * p.ldq + head * D;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (16 * s + 8 * half + 8 <= D) v = hv_ld16(qrow + 16 * s + 8 * half);
#ifndef HV_EMU
            // pin the value here: hipcc then waits for these ordinary loads BEFORE the first LDS-DMA is issued; left to
            // itself it puts `s_waitcnt vmcnt(0)` at their first use inside the tile loop, which drains the DMA ring every tile
            asm volatile("" : "+v"(v)::"memory");
#endif
            qf[b][s] = hv_as_bf16x8(v);
        }
    }

    f32x16 oacc[NQB][MT];
    float mrun[NQB], lrun[NQB];
#pragma unroll
    for (int b = 0; b < NQB; ++b) {
        mrun[b] = -INFINITY;
        lrun[b] = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[b][mt][r] = 0.f;
    }
    const float c2 = p.scale * 1.44269504089f;

    // fragment addresses inside a stage (constant over the tiles)
    const int krow = (col & ~0xc) | ((col & 4) << 1) | ((col & 8) >> 1);          // kappa(col): bits 2 and 3 swapped
    const int k_off = krow * G::KROW + hw * D * 2 + half * 16;                     // + 32 s
    const int tq = lane & 15, gq = lane >> 4;
    const int v_off = G::KBYTES + (8 * (gq >> 1) + (tq >> 2)) * G::VROW + (hw * D + 16 * (gq & 1) + 4 * (tq & 3)) * 2;
    // + (16 j + 4 u) * VROW + 64 mt

    // Scheduling hints: hipcc clusters the MFMAs of a region and lets the VALU work follow, but an in-order wave that has
    // just issued an MFMA can only issue the next one 32 cycles later -- program order must alternate "one MFMA, a few VALU"
    // for the two pipes to overlap inside ONE wave (the two waves of a SIMD run in lockstep between the barriers).
#ifndef HV_EMU
#define HV_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#else
#define HV_SGB(mask, n)
#endif
    constexpr int SGB_MFMA = 0x8, SGB_VALU = 0x402;  // VALU | TRANS

    // S^T = K.Q^T of one tile for query block b: sacc[r] = score of query `col`, key 16 (r >> 3) + 8 half + (r & 7)
    auto load_k = [&](int ti, bf16x8(&kf)[KS]) __attribute__((always_inline)) {
        const unsigned char* kst = smem + (ti % G::NST) * G::STAGE + k_off;
#pragma unroll
        for (int s = 0; s < KS; ++s) kf[s] = hv_as_bf16x8(hv_ld16(kst + 32 * s));
    };
    auto scores = [&](int b, const bf16x8(&kf)[KS], f32x16& sacc) __attribute__((always_inline)) {
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[b][0], z, 0, 0, 0);  // C = inline constant 0
#pragma unroll
        for (int s = 1; s < KS; ++s)
            if (!(HV_A2_DBG & 8)) sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s], qf[b][s], sacc, 0, 0, 0);
    };
    // ragged end of a source: keys past L score -inf (wave-uniform test; only the last tile of a source can be ragged)
    auto mask_tail = [&](int ti, f32x16(&sacc)[NQB]) __attribute__((always_inline)) {
        const bool bank = ti >= T1;
        const int kv0 = (bank ? ti - T1 : ti) * TK;
        const int L = bank ? p.L2 : p.L1;
        if (kv0 + TK > L) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (kv0 + 16 * (r >> 3) + 8 * half + (r & 7) >= L) {
#pragma unroll
                    for (int b = 0; b < NQB; ++b) sacc[b][r] = -INFINITY;
                }
            }
        }
    };
    // exp / row sum / bf16 pack of one query block against its (already updated) running maximum
    auto probs = [&](int b, const f32x16& sc, bf16x8(&pf)[2]) __attribute__((always_inline)) {
        const float mnew = mrun[b];
        float pv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) pv[r] = (HV_A2_DBG & 2) ? sc[r] : __builtin_amdgcn_exp2f(sc[r] * c2 - mnew);
        if (HV_A2_DBG & 2) {
            pf[0] = hv_as_bf16x8(u32x4{(unsigned)pv[0], (unsigned)pv[1], (unsigned)pv[2], (unsigned)pv[3]});
            pf[1] = hv_as_bf16x8(u32x4{(unsigned)pv[4], (unsigned)pv[5], (unsigned)pv[6], (unsigned)pv[7]});
            lrun[b] += pv[8];
            return;
        }
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 4) psum += (pv[r] + pv[r + 1]) + (pv[r + 2] + pv[r + 3]);
        lrun[b] += psum;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const u32x4 w = {hv_pack2(pv[8 * j], pv[8 * j + 1]), hv_pack2(pv[8 * j + 2], pv[8 * j + 3]),
                             hv_pack2(pv[8 * j + 4], pv[8 * j + 5]), hv_pack2(pv[8 * j + 6], pv[8 * j + 7])};
            pf[j] = hv_as_bf16x8(w);
        }
    };
    // One pipeline step.  The score registers of a query block are free as soon as its probabilities are packed, so S^T of
    // tile ti + 1 is computed straight back into them: block 0's S^T MFMAs run under block 1's exp / sum / pack VALU work,
    // and the MFMA tail of the step (S^T of block 1, then every P.V product) runs under the NEXT step's wait / barrier /
    // DMA issue / row maxima.  One score buffer, no register copies, one barrier per tile.
    auto step = [&](int ti, f32x16(&sacc)[NQB]) __attribute__((always_inline)) {
        // groups up to ti + 1 have landed for this wave (group ti + 2 may stay in flight) ...
        HV_TRACE(1);
        if (HV_A2_DBG & 1) hv_vm_wait<0>();
        else if (my_inst == G::MAXI) hv_vm_wait<G::MAXI>();
        else hv_vm_wait<G::MAXI - 1>();
        HV_TRACE(2);
        if (!(HV_A2_DBG & 16)) hv_barrier_raw();  // ... and for everybody else; all waves are done with tile ti - 1
        HV_TRACE(3);
        // group ti + 3 goes to the K slot of tile ti and the V slot of tile ti - 1.  Its instructions are spread over the
        // step: issued as one burst by all 8 waves right after the barrier they queue behind each other in the CU's load
        // path (~230 cycles per instruction, measured with attn2_trace) and the issuing waves stall for it
        // An LDS-DMA instruction stalls its wave for ~200-350 cycles here however it is placed.  The two waves of a SIMD
        // (w, w + 4) therefore take their turn at different points of the step: while one sits in its DMA issue the other
        // has the SIMD's VALU and matrix pipe to itself.
        const bool early = wave < 4;
        if (!(HV_A2_DBG & 1)) {
            issue_prepare(ti + 3);
            if (early) {
#pragma unroll
                for (int n = 0; n < G::MAXI; ++n) issue_part(ti + 3, n);
            }
        }
        HV_TRACE(4);
        mask_tail(ti, sacc);
        // ---- running maxima of both query blocks, lazy rescale (wave-uniform branch, rare after the first tiles)
        bool grew = false;
        float alpha[NQB];
#pragma unroll
        for (int b = 0; b < NQB; ++b) {
            float mx = fmaxf(fmaxf(sacc[b][0], sacc[b][1]), sacc[b][2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, sacc[b][r]), sacc[b][r + 1]);
            mx = fmaxf(mx, sacc[b][15]);
            mx = fmaxf(mx, hv_swap32(mx));
            const float mold = mrun[b];
            const float mnew = fmaxf(mold, mx * c2);
            mrun[b] = mnew;
            alpha[b] = mold - mnew;
            grew = grew || (mnew > mold);
        }
        if (__any(grew)) {
#pragma unroll
            for (int b = 0; b < NQB; ++b) {
                const float a = __builtin_amdgcn_exp2f(alpha[b]);
                lrun[b] *= a;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[b][mt][r] *= a;
            }
        }
        HV_TRACE(5);
        // V tile ti travelled in group ti + 1; one base address, immediate offsets
        const auto vbase = hv_lds_addr(smem + (ti % G::NST) * G::STAGE + v_off);
        bf16x4 vlo[MT][2], vhi[MT][2];
        hv_static_for<MT * 2>([&](auto ic) {
            constexpr int mt = decltype(ic)::value / 2, j = decltype(ic)::value % 2;
            hv_lds_tr4_issue_off<(16 * j) * G::VROW + 64 * mt>(vlo[mt][j], vbase);
            hv_lds_tr4_issue_off<(16 * j + 4) * G::VROW + 64 * mt>(vhi[mt][j], vbase);
        });
        bf16x8 pf[NQB][2];
        if constexpr (NQB > 1) {
            bf16x8 kf[KS];
            load_k(ti + 1, kf);  // past the last tile: the clamped duplicate of the last K tile (result unused)
#pragma unroll
            for (int b = 0; b < NQB; ++b) {
                probs(b, sacc[b], pf[b]);
                if (b > 0) {  // block b - 1's next scores under block b's probabilities
                    scores(b - 1, kf, sacc[b - 1]);
#pragma unroll
                    for (int i = 0; i < KS; ++i) {
                        HV_SGB(SGB_MFMA, 1);
                        HV_SGB(SGB_VALU, (56 + KS - 1) / KS);
                    }
                }
            }
            scores(NQB - 1, kf, sacc[NQB - 1]);
        } else {  // one query block: K fragments streamed through the MFMA chain (d = 160: 10 fragments would not fit)
            probs(0, sacc[0], pf[0]);
            const unsigned char* kst = smem + ((ti + 1) % G::NST) * G::STAGE + k_off;
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hv_as_bf16x8(hv_ld16(kst)), qf[0][0], z, 0, 0, 0);
#pragma unroll
            for (int s = 1; s < KS; ++s)
                sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hv_as_bf16x8(hv_ld16(kst + 32 * s)), qf[0][s], sacc[0], 0, 0, 0);
        }
        if (!(HV_A2_DBG & 1) && !early) {
#pragma unroll
            for (int n = 0; n < G::MAXI; ++n) issue_part(ti + 3, n);
        }
        hv_lds_tr4_wait();
        HV_TRACE(6);
#pragma unroll
        for (int b = 0; b < NQB; ++b)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bf16x4 lo = vlo[mt][j], hi = vhi[mt][j];
                    const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    if (!(HV_A2_DBG & 4) || lrun[0] == 1.2345f)
                        oacc[b][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[b][j], oacc[b][mt], 0, 0, 0);
                }
        HV_TRACE(7);
    };

    issue(0);
    issue(1);
    issue(2);
    if (my_inst == G::MAXI) hv_vm_wait<2 * G::MAXI>();
    else hv_vm_wait<2 * (G::MAXI - 1)>();
    hv_barrier_raw();
    f32x16 sacc[NQB];
    {
        bf16x8 kf[KS];
        load_k(0, kf);
#pragma unroll
        for (int b = 0; b < NQB; ++b) scores(b, kf, sacc[b]);
    }
    for (int ti = 0; ti < ntiles; ++ti) step(ti, sacc);
    hv_vm_wait<0>();  // the clamped look-ahead groups are still landing

    // ---- normalise and store: lane owns query `col`, channels 32 mt + 8 (r >> 2) + 4 half + (r & 3)
#pragma unroll
    for (int b = 0; b < NQB; ++b) {
        const float l = lrun[b] + hv_swap32(lrun[b]);
        const float inv = 1.0f / l;
        const int q = q_wave + 32 * b + col;
        if (q >= p.Lq) continue;
        bf16_t* dst = p.O + ((long)img * p.Lq + q) * p.ldo + head * D;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r1 = 0; r1 < 4; ++r1) {
                const int d = 32 * mt + 8 * r1 + 4 * half;
                if (d < D) {
                    const u32x2 o = {hv_pack2(oacc[b][mt][4 * r1] * inv, oacc[b][mt][4 * r1 + 1] * inv),
                                     hv_pack2(oacc[b][mt][4 * r1 + 2] * inv, oacc[b][mt][4 * r1 + 3] * inv)};
                    hv_st8(dst + d, o);
                }
            }
    }
}

template <int D>
static inline void hv_attention2_launch_t(const hv_attention_params& p, hipStream_t stream) {
    using G = HvAttn2Geom<D>;
    const int total = ((p.Lq + G::BQ - 1) / G::BQ) * (p.heads / G::HG) * p.n_images;
    const int grid = ((total + 7) / 8) * 8;
    hv_note("hv_attention2_kernel<%d> | n=%d heads=%d D=%d Lq=%d L1=%d L2=%d bank=%d", D, p.n_images, p.heads, D, p.Lq, p.L1,
            p.L2, p.bank_sel != nullptr && p.L2 > 0);
    hv_launch(hv_attention2_kernel<D>, dim3(grid), dim3(512), stream, p);
}

// row-major V form of hv_attention (hv_attention_params.v_row_major = 1)
static inline int hv_attention2_launch(const hv_attention_params& p, hipStream_t stream) {
    if (p.L1 <= 0 || p.L2 < 0 || p.Lq <= 0 || p.heads * p.D % 160 != 0) return -1;
    if (p.ldq % 8 || p.ldk % 8 || p.ldvt % 8 || p.ldo % 4) return -1;
    if (p.L2 > 0 && p.bank_sel != nullptr && (!p.K2 || !p.Vt2 || p.ldk2 % 8 || p.ldvt2 % 8)) return -1;
    switch (p.D) {
        case 40: hv_attention2_launch_t<40>(p, stream); break;
        case 80: hv_attention2_launch_t<80>(p, stream); break;
        case 160: hv_attention2_launch_t<160>(p, stream); break;
        default: return -2;
    }
    return 0;
}
