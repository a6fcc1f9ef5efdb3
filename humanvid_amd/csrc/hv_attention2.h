// hv_attention2.h -- spatial self-attention with reference-bank keys, round-2 kernel (gfx950).
//
// Same semantics as hv_attention.h (diffusers SDPA as called by the read-mode patched transformer block,
// /root/reference/src/models/mutual_self_attention.py:147-186), different data path.  What the round-1 kernel lost its
// time on (profiles/README.md, attn_trace): per 64-key tile 1365 of 4700 cycles ISSUING global loads (a head's K rows are
// 80-byte pieces of 640-byte token rows), 611 cycles storing the tile to LDS from registers, and V needed a transposed
// copy that the QKV GEMM wrote with 2-byte scattered stores.  Here:
//
//  * one workgroup = 8 waves = a 320-channel slice of ALL its heads (8 / 4 / 2 heads for d = 40 / 80 / 160, one wave per
//    head and query group), so K and V are consumed as whole 640-byte row pieces straight out of the row-major
//    [token][q | k | v] tensor the QKV GEMM writes: no head-major re-layout, no V^T tensor;
//  * K / V tiles (32 keys) go global -> LDS by LDS-DMA (global_load_lds, 1 KiB per wave-instruction) into a 3-stage
//    ring, two tiles in flight across ONE barrier per tile with counted vmcnt; the LDS image is built by the per-lane
//    SOURCE addresses: K rows 41 chunks (656 B) apart, V rows 44 chunks (704 B) apart -- the extra chunk positions are
//    dummy loads -- which makes both fragment reads bank-conflict free;
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

template <int D>
struct HvAttn2Geom {
    static constexpr int HG = 320 / D;             // heads per workgroup
    static constexpr int QG = 8 / HG;              // query groups (waves per head)
    static constexpr int NQB = D == 40 ? 2 : 1;    // 32-query blocks per wave
    static constexpr int BQ = 32 * NQB * QG;       // queries per workgroup: 64 / 64 / 128
    static constexpr int TK = 32;                  // keys per tile
    static constexpr int KS = (D + 15) / 16;       // 16-deep steps of the QK^T reduction: 3 / 5 / 10
    static constexpr int MT = (D + 31) / 32;       // 32-row tiles of O^T: 2 / 3 / 5
    static constexpr int KCH = 41, VCH = 44;       // 16-byte chunk positions per LDS row (40 real + dummies)
    static constexpr int KROW = KCH * 16, VROW = VCH * 16;
    static constexpr int KINST = (TK * KCH + 63) / 64, VINST = (TK * VCH + 63) / 64;  // wave-instructions per tile: 21 + 22
    static constexpr int NINST = KINST + VINST;
    static constexpr int KBYTES = KINST * 1024, STAGE = NINST * 1024;
    static constexpr int NST = 3;
    static constexpr int MAXI = (NINST + 7) / 8;   // DMA instructions per wave and tile (6)
};

template <int D>
__global__ __launch_bounds__(512, 2) void hv_attention2_kernel(hv_attention_params p) {
    using G = HvAttn2Geom<D>;
    constexpr int HG = G::HG, NQB = G::NQB, KS = G::KS, MT = G::MT, TK = G::TK;
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::NST * G::STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int hw = wave % HG, qg = wave / HG;

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
    const int c0 = hgi * 320;                      // first channel of this head group
    const int head = hgi * HG + hw;

    // ---- DMA plan of this wave: instruction i = wave + 8 n fills LDS chunks [64 i, 64 i + 64) of a stage
    int dma_row[G::MAXI], dma_col[G::MAXI];        // key row inside the tile, byte offset inside the 640-byte row piece
#pragma unroll
    for (int n = 0; n < G::MAXI; ++n) {
        const int i = wave + 8 * n;
        const bool isk = i < G::KINST;
        const int j = (isk ? i : i - G::KINST) * 64 + lane;
        const int per = isk ? G::KCH : G::VCH;
        dma_row[n] = min(j / per, TK - 1);
        dma_col[n] = min(j % per, 39) * 16;
    }
    auto issue = [&](int ti) __attribute__((always_inline)) {
        const bool bank = ti >= T1;
        const int kv0 = (bank ? ti - T1 : ti) * TK;
        const int L = bank ? p.L2 : p.L1;
        const long rowbase = bank ? (long)sel * p.L2 : (long)img * p.L1;
        const char* kb = reinterpret_cast<const char*>(bank ? p.K2 : p.K) + c0 * 2;
        const char* vb = reinterpret_cast<const char*>(bank ? p.Vt2 : p.Vt) + c0 * 2;
        const long ldk = (bank ? p.ldk2 : p.ldk) * 2, ldv = (bank ? p.ldvt2 : p.ldvt) * 2;
        unsigned char* stage = smem + (ti % G::NST) * G::STAGE;
#pragma unroll
        for (int n = 0; n < G::MAXI; ++n) {
            const int i = wave + 8 * n;
            if (i < G::NINST) {
                const long row = rowbase + min(kv0 + dma_row[n], L - 1);
                const char* src = i < G::KINST ? kb + row * ldk + dma_col[n] : vb + row * ldv + dma_col[n];
                hv_glds16(src, stage + i * 1024);
            }
        }
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

    issue(0);
    if (ntiles > 1) issue(1);
    for (int ti = 0; ti < ntiles; ++ti) {
        // this wave's pieces of tile ti have landed (tile ti + 1 may stay in flight) ...
        if (ti + 1 < ntiles) {
            if (my_inst == G::MAXI) hv_vm_wait<G::MAXI>();
            else hv_vm_wait<G::MAXI - 1>();
        } else {
            hv_vm_wait<0>();
        }
        hv_barrier_raw();  // ... and everybody else's; all waves are done reading tile ti - 1
        if (ti + 2 < ntiles) issue(ti + 2);  // into the stage of tile ti - 1
        const unsigned char* stage = smem + (ti % G::NST) * G::STAGE;

        // ---- S^T = K.Q^T : sacc[b][r] = score of query `col`, key 16 (r >> 3) + 8 half + (r & 7)
        f32x16 sacc[NQB];
#pragma unroll
        for (int b = 0; b < NQB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[b][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 kf = hv_as_bf16x8(hv_ld16(stage + k_off + 32 * s));
#pragma unroll
            for (int b = 0; b < NQB; ++b) sacc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[b][s], sacc[b], 0, 0, 0);
        }
        // ragged end of a source: keys past L score -inf (wave-uniform test, last tile of each source only)
        {
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
        }
        // ---- online softmax (exp2 domain); P^T fragments for the two 16-key steps of V^T.P^T
        bf16x8 pf[NQB][2];
#pragma unroll
        for (int b = 0; b < NQB; ++b) {
            float mx = fmaxf(sacc[b][0], sacc[b][1]);
#pragma unroll
            for (int r = 2; r < 16; r += 2) mx = fmaxf(mx, fmaxf(sacc[b][r], sacc[b][r + 1]));
            mx = fmaxf(mx, hv_swap32(mx));
            const float mold = mrun[b];
            const float mnew = fmaxf(mold, mx * c2);
            mrun[b] = mnew;
            if (__any(mnew > mold)) {  // some query of this wave raised its maximum: rescale the accumulators
                const float alpha = __builtin_amdgcn_exp2f(mold - mnew);
                lrun[b] *= alpha;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[b][mt][r] *= alpha;
            }
            float pv[16];
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pv[r] = __builtin_amdgcn_exp2f(sacc[b][r] * c2 - mnew);
                psum += pv[r];
            }
            lrun[b] += psum;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const u32x4 w = {hv_pack2(pv[8 * j], pv[8 * j + 1]), hv_pack2(pv[8 * j + 2], pv[8 * j + 3]),
                                 hv_pack2(pv[8 * j + 4], pv[8 * j + 5]), hv_pack2(pv[8 * j + 6], pv[8 * j + 7])};
                pf[b][j] = hv_as_bf16x8(w);
            }
        }
        // ---- O^T += V^T.P^T : V^T fragments by transposing LDS reads of the row-major V tile
        bf16x4 vlo[MT][2], vhi[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned char* va = stage + v_off + (16 * j) * G::VROW + 64 * mt;
                hv_lds_tr4_issue(vlo[mt][j], va);
                hv_lds_tr4_issue(vhi[mt][j], va + 4 * G::VROW);
            }
        hv_lds_tr4_wait();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x4 lo = vlo[mt][j], hi = vhi[mt][j];
                const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int b = 0; b < NQB; ++b)
                    oacc[b][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[b][j], oacc[b][mt], 0, 0, 0);
            }
    }

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
    if (p.L1 <= 0 || p.L2 < 0 || p.Lq <= 0 || p.heads * p.D % 320 != 0) return -1;
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
