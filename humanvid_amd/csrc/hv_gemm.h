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

// Channel permutation of a wave's 64-channel block (round 3, LDS-DMA kernel, plain / residual / LayerNorm-fold outputs).
// The MFMA gives a lane 4 consecutive rows of the A operand per 16x16 fragment; with the natural assignment "fragment nf,
// row i <-> channel 16 nf + i" a lane owns 4-channel pieces 16 channels apart: 8-byte residual loads and output stores, 16
// token rows x 32 bytes per wave-instruction -- the epilogue of a K = 320 tile is store-ISSUE-bound (7400 of ~29 000 cycles,
// profiles/r02_gemm_trace.txt).  Assigning  fragment nf, row i <-> channel 32 (nf >> 1) + 8 (i >> 2) + 4 (nf & 1) + (i & 3)
// instead makes the fragment pair (2j, 2j+1) of a lane 8 CONSECUTIVE channels 32 j + 8 quad + 0..7: 16-byte loads and
// stores, 64 contiguous bytes per token and instruction, half the instructions.  It is only a different LDS row per
// (fragment, lane) in the W-fragment reads; the W tile gets its own XOR swizzle so that those reads stay conflict-free
// (checked per ds_read_b128 lane group: the X swizzle would make them 2-way).
HV_DEV int hv_perm_row(int f, int i) { return 32 * (f >> 1) + 8 * (i >> 2) + 4 * (f & 1) + (i & 3); }
// GEGLU (weight rows packed in 16-row [h | g] blocks, packing.geglu_row_order): fragment nf, row i = 4 q + e <-> packed row
// of  h (nf even) / g (nf odd)  of the wave's output channel o = 8 q + e + 4 (nf >> 1)  -- a lane's two gated results per
// row fragment are then the 8 consecutive output channels 8 quad + 0..7: one 16-byte store instead of two 8-byte ones.
HV_DEV int hv_perm_row_geglu(int f, int i) {
    const int o = 8 * (i >> 2) + (i & 3) + 4 * (f >> 1);
    return 32 * (o >> 4) + 16 * (f & 1) + (o & 15);
}
// XOR swizzle of the W tile under either permuted assignment: conflict-free for every ds_read_b128 lane group of both
// (searched over bit-parity swizzles; the X tile keeps hv_swz)
HV_DEV int hv_wperm_swizzle(int row) { return (row & 1) | (((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2); }
HV_DEV int hv_swz_wperm(int row, int chunk) {  // BK = 64: 128-byte rows, 8 chunks
    return row * 128 + ((chunk ^ hv_wperm_swizzle(row)) << 4);
}

// GELU (exact form, F.gelu default: src/models/attention.py GEGLU.gelu) of four gate values times four values:
//   gelu(x) = x Phi(x) = max(x, 0) - |x| / 2 * erfc(|x| / sqrt 2),   erfc(|x| / sqrt 2) = 2^(u q(u)), u = min(|x|, 4 sqrt 2)
// with q a degree-5 polynomial fitted (weighted minimax, oracle-side check in tests/kernel_cases.py case_gemm_geglu) to
// log2 erfc on [0, 4]: |gelu error| < 6e-7 absolute, < 5e-4 of max(|gelu|, 1e-3) -- below the bf16 rounding of the output
// (2^-9) everywhere.  One transcendental (v_exp_f32) and the polynomial in packed fp32 (v_pk_fma_f32: two elements per
// instruction), against one rcp + one exp2 + 12 scalar VALU per element of the Abramowitz-Stegun 7.1.26 form of rounds 1-2:
// the GEGLU epilogue of a 256 x 256 tile was VALU-bound on it (5000 of the tile's 33 000 cycles at K = 320).
HV_DEV f32x2 hv_gelu_times2(f32x2 x, f32x2 h) {
    const f32x2 u = {fminf(fabsf(x[0]), 5.656854249f), fminf(fabsf(x[1]), 5.656854249f)};
    f32x2 q = u * 1.775511355e-05f + -6.477572639e-04f;
    q = q * u + 7.724042040e-03f;
    q = q * u + -5.292673633e-02f;
    q = q * u + -4.590827375e-01f;
    q = q * u + -1.151116856e+00f;
    q = q * u;
    const f32x2 e = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
    const f32x2 pos = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
    return (pos - (u * 0.5f) * e) * h;
}
HV_DEV f32x4 hv_gelu_times(f32x4 x, f32x4 h) {
    const f32x2 a = hv_gelu_times2(f32x2{x[0], x[1]}, f32x2{h[0], h[1]}), b = hv_gelu_times2(f32x2{x[2], x[3]}, f32x2{h[2], h[3]});
    return f32x4{a[0], a[1], b[0], b[1]};
}

constexpr int HV_GEMM_EPI_G = 2;  // row fragments per load group of the fast epilogues (the register budget of the 256 x 256 kernel)
#ifndef HV_EPI_G4
#define HV_EPI_G4 0  // 1: the permuted epilogue of the 128 x 128 kernel (4 row fragments) requests all its per-row terms in ONE group
#endif

// ---- epilogue of one wave's (16*NMF) x 64 sub-tile: lane owns token m (column of the MFMA tile) and 4
//      consecutive channels n; m_base / n_base are the sub-tile origin.
// * All global loads of a phase are issued back-to-back before the first use: with LDS-DMA loads in
//   flight hipcc drains the whole VMEM queue (vmcnt(0)) at every ordinary load's first use, so a
//   load->use->load->use epilogue serialises dozens of L2 round trips per tile.
// * MODE picks a lean instantiation for the three hot output forms (1: bf16 row-major, 2: the same with
//   a transposed tail = the QKV projection, 3: GEGLU); MODE 0 handles everything (activation, fp32
//   output).  The fully general body is ~100 KB of code per kernel and ran out of the instruction
//   cache once per tile (measured: 0.29 of 0.61 ms of the level-0 QKV GEMM with the stores disabled).
template <int NMF, int MODE>
HV_DEV void hv_gemm_epilogue_t(const HvGemmParams& p, f32x4 (&acc)[4][NMF], int m_base, int n_base, int r16, int quad) {
    const bool geglu = MODE == 3 ? true : (MODE == 0 ? p.geglu != 0 : false);
    const bool has_t = MODE == 2 ? true : (MODE == 0 ? p.Yt != nullptr : false);
    const bool out_f32 = MODE == 0 ? p.out_f32 != 0 : false;
    const int out_act = MODE == 0 ? p.out_act : HV_ACT_NONE;
    const bool ln = p.row_rstd != nullptr;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // Loads use clamped (always valid) addresses and sit only under wave-uniform pointer tests, so that
    // hipcc keeps each group back-to-back and waits once; ragged edges are masked at the stores.
    int nc[4];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) nc[nf] = min(n_base + 16 * nf + 4 * quad, p.N - 4);
    const bool two_rows = p.pe != nullptr && p.rowvec != nullptr;  // both per-row tables: rare, slow path below
#pragma unroll
    for (int mf = 0; mf < NMF; ++mf) {
        const int m = m_base + 16 * mf + r16;
        const int mc = min(m, p.M - 1);
        int mo = mc;  // output / residual row (general instantiation only: optional transpose of the two outer row axes)
        if (MODE == 0 && p.perm_p > 0) {
            const int blk = mc / p.perm_p, pp = mc - blk * p.perm_p;
            const int px = blk / p.perm_y, py = blk - px * p.perm_y;
            mo = (py * p.perm_x + px) * p.perm_p + pp;
        }
        // phase 1: issue every load of this row fragment (the per-column vectors are L1 hits after the
        // first fragment; re-loading them keeps 48 VGPRs free across the fragments)
        f32x4 bias4[4], cs4[4], row4[4];
        u32x2 res2[4];
        float mean = 0.f, rstd = 1.f;
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) bias4[nf] = cs4[nf] = row4[nf] = zero4, res2[nf] = u32x2{0u, 0u};
        {
            if (p.bias != nullptr) {
#pragma unroll
                for (int nf = 0; nf < 4; ++nf) bias4[nf] = *reinterpret_cast<const f32x4*>(p.bias + nc[nf]);
            }
            if (ln) {
#pragma unroll
                for (int nf = 0; nf < 4; ++nf) cs4[nf] = *reinterpret_cast<const f32x4*>(p.colsum + nc[nf]);
                mean = p.row_mean[mc];
                rstd = p.row_rstd[mc];
            }
        }
        if (p.pe != nullptr) {
            const float* pe_row = p.pe + (long)((mc / p.pe_period) % p.pe_frames) * p.N;
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) row4[nf] = *reinterpret_cast<const f32x4*>(pe_row + nc[nf]);
        } else if (p.rowvec != nullptr) {
            const float* rv_row = p.rowvec + (long)(mc / p.rowvec_period) * p.N;
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) row4[nf] = *reinterpret_cast<const f32x4*>(rv_row + nc[nf]);
        }
        if (p.residual != nullptr) {
            const bf16_t* res_row = p.residual + (long)mo * p.ldr;
            if (!geglu) {
#pragma unroll
                for (int nf = 0; nf < 4; ++nf) res2[nf] = hv_ld8(res_row + nc[nf]);
            } else {
#pragma unroll
                for (int nf = 1; nf < 4; nf += 2)
                    res2[nf] = hv_ld8(res_row + min(((n_base + 16 * (nf - 1)) >> 1) + 4 * quad, (p.N >> 1) - 4));
            }
        }
        if (two_rows) {
            const float* rv_row = p.rowvec + (long)(mc / p.rowvec_period) * p.N;
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) row4[nf] += *reinterpret_cast<const f32x4*>(rv_row + nc[nf]);
        }
        // phase 2: arithmetic and stores
        char* yrow = reinterpret_cast<char*>(p.Y) + (long)(MODE == 0 ? (m < p.M ? mo : m) : m) * p.ldy * (out_f32 ? 4 : 2);
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const int n = n_base + 16 * nf + 4 * quad;
            const bool valid = m < p.M && n < p.N;
            f32x4 v = acc[nf][mf];
            if (ln) v = rstd * (v - mean * cs4[nf]);
            v += bias4[nf] + row4[nf];
            const f32x4 rres = {hv_bf2f((bf16_t)(res2[nf][0] & 0xffff)), hv_bf2f((bf16_t)(res2[nf][0] >> 16)),
                                hv_bf2f((bf16_t)(res2[nf][1] & 0xffff)), hv_bf2f((bf16_t)(res2[nf][1] >> 16))};
            if (geglu) {
                // packed weight rows: [16 x h | 16 x g] blocks -> fragment pairs (even nf: h, odd nf: g)
                acc[nf][mf] = v;
                if ((nf & 1) == 0) continue;
                const int no = ((n_base + 16 * (nf - 1)) >> 1) + 4 * quad;
                v = hv_gelu_times(v, acc[nf - 1][mf]);
                v += rres;
                u32x2 o = {hv_pack2(v[0], v[1]), hv_pack2(v[2], v[3])};
                if (valid) hv_st8_stream(yrow + 2 * no, o);
                continue;
            }
            v += rres;
            if (out_act != HV_ACT_NONE) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = hv_act(v[r], out_act);
            }
            if (!valid) continue;
            if (has_t && n >= p.n_split) {
                bf16_t* yt = p.Yt + (long)(n - p.n_split) * p.ldyt + m;
#pragma unroll
                for (int r = 0; r < 4; ++r) yt[(long)r * p.ldyt] = hv_f2bf(v[r]);
            } else if (out_f32) {
                *reinterpret_cast<f32x4*>(yrow + 4 * n) = v;
            } else {
                u32x2 o = {hv_pack2(v[0], v[1]), hv_pack2(v[2], v[3])};
                hv_st8_stream(yrow + 2 * n, o);
            }
        }
    }
}

// ---- fast epilogue for the hot output forms (round 2).  What was wrong with the one above, measured with gemm_trace and the
// timing-only builds: it costs ~20 000 cycles per K = 320 tile against ~24 000 for the tile's ten k-steps, and 0.18 of the
// 0.47 ms of the level-0 QKV GEMM remain when its stores are compiled out.  The stores of row fragment mf sit in front of the
// loads of fragment mf+1; hipcc may not hoist those loads (Y can alias the residual -- the residual stream IS updated in
// place), and with LDS-DMA in flight every wait for them is a vmcnt(0), which on gfx9 also waits for the just-issued stores
// to be acknowledged: eight serial store round trips per tile.  Here NO store precedes a load: the per-column vectors are
// loaded once, the per-row terms (LayerNorm mean / rstd, residual) in groups of four row fragments, the packed results take
// the place of the accumulators they consume, and all stores of the tile go out back to back at the end and drain under the
// next tile's k-steps.  Everything is compile-time selected (LN fold, residual, output form) -- straight-line code, ragged
// edges by clamped loads and masked stores.  A per-row table (positional encoding or per-batch vector) is supported when
// one table row covers the wave's rows (always true at the levels that matter: tokens per image are multiples of 128);
// other tiles, fp32 output, activations and the row permutation take the general epilogue.
// OUT: 0 = bf16 row-major, 1 = the same with the transposed tail (QKV projection), 2 = GEGLU.
template <int NMF, bool LN, bool RES, int OUT>
HV_DEV void hv_gemm_epilogue_fast(const HvGemmParams& p, f32x4 (&acc)[4][NMF], int m_base, int n_base, int r16, int quad,
                                  const float* tab_row) {
    static_assert(!(RES && OUT != 0), "residual only with the plain output form");
    constexpr int G = NMF < HV_GEMM_EPI_G ? NMF : HV_GEMM_EPI_G;  // row fragments per load group
    constexpr int NO = OUT == 2 ? 2 : 4;  // packed 4-channel outputs per row fragment
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    int nc[4];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) nc[nf] = min(n_base + 16 * nf + 4 * quad, p.N - 4);
    f32x4 add4[4], cs4[4];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) add4[nf] = cs4[nf] = zero4;
    // every address below is a wave-uniform base + a 32-bit byte offset (saddr form: one address register per load)
    unsigned nb[4];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) nb[nf] = 4u * (unsigned)nc[nf];
    auto ld4 = [&](const float* base, unsigned byte_ofs) __attribute__((always_inline)) {
        return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + byte_ofs);
    };
    if (p.bias != nullptr) {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) add4[nf] = ld4(p.bias, nb[nf]);
    }
    if (LN) {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) cs4[nf] = ld4(p.colsum, nb[nf]);
    }
    if (tab_row != nullptr) {
        f32x4 t4[4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) t4[nf] = ld4(tab_row, nb[nf]);
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) add4[nf] += t4[nf];
    }
    u32x2 outp[NMF][NO];
#pragma unroll
    for (int g = 0; g < NMF; g += G) {
        float mean[G], rstd[G];
        u32x2 res2[G][RES ? 4 : 1];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int mc = min(m_base + 16 * (g + j) + r16, p.M - 1);
            if (LN) {
                mean[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.row_mean) + 4u * (unsigned)mc);
                rstd[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.row_rstd) + 4u * (unsigned)mc);
            }
            if (RES) {
                // 32-bit byte offsets from the wave-uniform base (span checked by hv_gemm_launch): half the address registers
                const unsigned ro = (unsigned)mc * (unsigned)p.ldr * 2u;
#pragma unroll
                for (int nf = 0; nf < 4; ++nf)
                    res2[j][nf] = hv_ld8(reinterpret_cast<const char*>(p.residual) + (ro + 2u * (unsigned)nc[nf]));
            }
        }
#if !defined(HV_EMU)
        __builtin_amdgcn_sched_barrier(0);  // the group's loads stay together, ahead of its arithmetic
#endif
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int mf = g + j;
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                f32x4 v = acc[nf][mf];
                if (LN) v = rstd[j] * (v - mean[j] * cs4[nf]);
                v += add4[nf];
                if (OUT == 2) {
                    // packed weight rows: [16 x h | 16 x g] blocks -> fragment pairs (even nf: h, odd nf: g)
                    if ((nf & 1) == 0) {
                        acc[nf][mf] = v;
                        continue;
                    }
                    v = hv_gelu_times(v, acc[nf - 1][mf]);
                    outp[mf][OUT == 2 ? (nf >> 1) : 0] = u32x2{hv_pack2(v[0], v[1]), hv_pack2(v[2], v[3])};
                } else {
                    if (RES) {
                        const u32x2 r2 = res2[j][RES ? nf : 0];
                        v += f32x4{hv_bf2f((bf16_t)(r2[0] & 0xffff)), hv_bf2f((bf16_t)(r2[0] >> 16)),
                                   hv_bf2f((bf16_t)(r2[1] & 0xffff)), hv_bf2f((bf16_t)(r2[1] >> 16))};
                    }
                    outp[mf][OUT == 2 ? 0 : nf] = u32x2{hv_pack2(v[0], v[1]), hv_pack2(v[2], v[3])};
                }
            }
        }
#if !defined(HV_EMU)
        __builtin_amdgcn_sched_barrier(0);  // ... and the next group's loads are not hoisted over it (register budget)
#endif
    }
#ifndef HV_EMU
    // Pin the packed results here.  Otherwise hipcc sinks the arithmetic of a fragment into its (edge-masked) store branch;
    // on the path that skips the store the fragment's loads are then never waited for, the register state that reaches the
    // k-loop header carries "load in flight into v[..]", and the first ds_read of the next k-step that reuses such a register
    // gets a compiler-inserted s_waitcnt vmcnt(0) -- in EVERY k-step, draining the LDS-DMA ring (seen in the .s of the first
    // version of this function: two vmcnt(0) between the k-loop's ds_reads).
#pragma unroll
    for (int mf = 0; mf < NMF; ++mf)
#pragma unroll
        for (int no = 0; no < NO; ++no) asm volatile("" : "+v"(outp[mf][no][0]), "+v"(outp[mf][no][1]));
    // ... and tell hipcc's wait-count tracker that no load is outstanding from here on (only this tile's stores and the
    // LDS-DMA of the next k-tiles follow): vmcnt(0), lgkmcnt / expcnt untouched.  No store has been issued yet, so this
    // waits for loads only.
    __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
    char* const yb = reinterpret_cast<char*>(p.Y);
    char* const ytb = reinterpret_cast<char*>(p.Yt);
#pragma unroll
    for (int mf = 0; mf < NMF; ++mf) {
        const int m = m_base + 16 * mf + r16;
        if (m >= p.M) continue;
        const unsigned yo = (unsigned)m * (unsigned)p.ldy * 2u;
#pragma unroll
        for (int no = 0; no < NO; ++no) {
            const u32x2 o = outp[mf][no];
            if (OUT == 2) {
                const int n = n_base + 32 * no + 4 * quad;  // h column of the [h|g] pair (the g column is 16 further)
                if (n + 16 < p.N) hv_st8_stream(yb + (yo + 2u * (unsigned)((n_base >> 1) + 16 * no + 4 * quad)), o);
            } else {
                const int n = n_base + 16 * no + 4 * quad;
                if (n >= p.N) continue;
                if (OUT == 1 && n >= p.n_split) {
                    const unsigned ts = (unsigned)p.ldyt * 2u, to = (unsigned)(n - p.n_split) * ts + 2u * (unsigned)m;
                    *reinterpret_cast<bf16_t*>(ytb + to) = (bf16_t)(o[0] & 0xffffu);
                    *reinterpret_cast<bf16_t*>(ytb + (to + ts)) = (bf16_t)(o[0] >> 16);
                    *reinterpret_cast<bf16_t*>(ytb + (to + 2 * ts)) = (bf16_t)(o[1] & 0xffffu);
                    *reinterpret_cast<bf16_t*>(ytb + (to + 3 * ts)) = (bf16_t)(o[1] >> 16);
                } else {
                    hv_st8_stream(yb + (yo + 2u * (unsigned)n), o);
                }
            }
        }
    }
}

// The same epilogue for the permuted channel assignment (hv_perm_row): plain bf16 row-major output with optional LayerNorm
// fold / residual; the fragment pair (2j, 2j+1) of a lane is 8 consecutive channels n_base + 32 j + 8 quad.  N % 8 == 0.
// STATS: additionally leaves the GroupNorm partial statistics of the stored tile in p.gn_part (per channel: sum and sum of
// squares over the wave's rows; see hv_gemm_params) -- in-lane over the row fragments, then over the 16 lanes of the DPP row.
// STATS = 2: LayerNorm partial statistics instead (p.ln_part): per ROW the sum and the sum of squares over the wave's 64
// columns -- in-lane over the lane's 16 channels, then over the four quads that share a row.
template <int NMF, bool LN, bool RES, int STATS = 0>
HV_DEV void hv_gemm_epilogue_fast_perm(const HvGemmParams& p, f32x4 (&acc)[4][NMF], int m_base, int n_base, int r16, int quad,
                                       const float* tab_row) {
    constexpr int G = (NMF <= 4 && HV_EPI_G4) ? NMF : (NMF < HV_GEMM_EPI_G ? NMF : HV_GEMM_EPI_G);  // row fragments per load group
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    int nc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) nc[j] = min(n_base + 32 * j + 8 * quad, p.N - 8);
    f32x4 add4[4], cs4[4];  // [2 j + h]: channels nc[j] + 4 h .. + 3
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) add4[nf] = cs4[nf] = zero4;
    auto ld4 = [&](const float* base, unsigned byte_ofs) __attribute__((always_inline)) {
        return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + byte_ofs);
    };
    if (p.bias != nullptr) {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) add4[nf] = ld4(p.bias, 4u * (unsigned)(nc[nf >> 1] + 4 * (nf & 1)));
    }
    if (LN) {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) cs4[nf] = ld4(p.colsum, 4u * (unsigned)(nc[nf >> 1] + 4 * (nf & 1)));
    }
    // Positional-encoding table whose period is not a multiple of the wave's row block (level 3: 96 tokens per frame under
    // 64-row blocks) but of 16: one table row per 16-row FRAGMENT, loaded with the fragment group's other per-row terms.
    // (LayerNorm-fold form only: the temporal QKV projection, motion_module.py:157-175 + PositionalEncoding :132-147.)
    constexpr bool FRAG_TAB_FORM = LN && !RES && STATS == 0 && NMF <= 4;  // (the 128 x 128 kernel: the 256 x 256 one has no registers for it)
    const bool tab_per_frag = FRAG_TAB_FORM && p.pe != nullptr && p.pe_period % (16 * NMF) != 0;
    if (tab_row != nullptr && !tab_per_frag) {
        f32x4 t4[4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) t4[nf] = ld4(tab_row, 4u * (unsigned)(nc[nf >> 1] + 4 * (nf & 1)));
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) add4[nf] += t4[nf];
    }
    // Row permutation of the output (p.perm_p > 0, hv_gemm_params: row (x perm_y + y) perm_p + pp is stored -- and the residual
    // read -- at (y perm_x + x) perm_p + pp: the frame-sharded motion module writes its all-to-all send layout directly).
    // The lane's rows are m_base + r16 + 16 mf: two divisions for the first, steps of 16 for the rest (perm_p >= 16: host).
    // Everything on the INPUT side (LayerNorm row statistics, per-row tables, the bounds test) keeps the logical row m;
    // everything on the output side (Y, residual, the LayerNorm partial statistics of the output) uses mo[mf].
    // (128 x 128 kernel only -- the 256 x 256 one has no registers for it: hv_gemm_choose sends row-permuted problems here)
    constexpr bool ROWPERM = NMF <= 4;
    int mo[NMF];
    {
        const int m0r = m_base + r16;
        if (ROWPERM && p.perm_p > 0) {
            int blk = m0r / p.perm_p, pp = m0r - blk * p.perm_p;
            int px = blk / p.perm_y, py = blk - px * p.perm_y;
#pragma unroll
            for (int mf = 0; mf < NMF; ++mf) {
                mo[mf] = (py * p.perm_x + px) * p.perm_p + pp;
                pp += 16;
                if (pp >= p.perm_p) {
                    pp -= p.perm_p;
                    if (++py == p.perm_y) py = 0, ++px;
                }
            }
        } else {
#pragma unroll
            for (int mf = 0; mf < NMF; ++mf) mo[mf] = m0r + 16 * mf;
        }
    }
    u32x4 outp[NMF][2];
    f32x4 gs[4], gq[4];  // STATS = 1: per-channel partial sums of this lane ([2 h + k]: channels nc[h] + 4 k .. + 3)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) gs[nf] = gq[nf] = zero4;
    float rs[NMF], rq[NMF];  // STATS = 2: per-row partial sums of this lane (16 of the wave's 64 columns)
#pragma unroll
    for (int mf = 0; mf < NMF; ++mf) rs[mf] = rq[mf] = 0.f;
#pragma unroll
    for (int g = 0; g < NMF; g += G) {
        float mean[G], rstd[G];
        u32x4 res4[G][RES ? 2 : 1];
        f32x4 tf4[FRAG_TAB_FORM ? G : 1][4];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int mc = min(m_base + 16 * (g + j) + r16, p.M - 1);
            if (LN) {
                mean[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.row_mean) + 4u * (unsigned)mc);
                rstd[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.row_rstd) + 4u * (unsigned)mc);
            }
            if constexpr (FRAG_TAB_FORM) {
#pragma unroll
                for (int nf = 0; nf < 4; ++nf) tf4[j][nf] = zero4;
                if (tab_per_frag) {  // (wave-uniform)
                    const int mrow = min(m_base + 16 * (g + j), p.M - 1);
#ifndef HV_EMU
                    const int fr = __builtin_amdgcn_readfirstlane((mrow / p.pe_period) % p.pe_frames);
#else
                    const int fr = (mrow / p.pe_period) % p.pe_frames;
#endif
                    const float* trow = p.pe + (long)fr * p.N;
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf) tf4[j][nf] = ld4(trow, 4u * (unsigned)(nc[nf >> 1] + 4 * (nf & 1)));
                }
            }
            if (RES) {
                // (rows beyond M: any valid row -- the clamped logical one -- their results are not stored)
                const unsigned ro = (unsigned)(m_base + 16 * (g + j) + r16 < p.M ? mo[g + j] : mc) * (unsigned)p.ldr * 2u;
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    res4[j][h] = hv_ld16(reinterpret_cast<const char*>(p.residual) + (ro + 2u * (unsigned)nc[h]));
            }
        }
#if !defined(HV_EMU)
        __builtin_amdgcn_sched_barrier(0);  // the group's loads stay together, ahead of its arithmetic
#endif
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int mf = g + j;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4 o;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int nf = 2 * h + k;
                    f32x4 v = acc[nf][mf];
                    if (LN) v = rstd[j] * (v - mean[j] * cs4[nf]);
                    v += add4[nf];
                    if constexpr (FRAG_TAB_FORM) v += tf4[j][nf];
                    if (RES) {
                        const unsigned r0 = res4[j][RES ? h : 0][2 * k], r1 = res4[j][RES ? h : 0][2 * k + 1];
                        v += f32x4{hv_bf2f((bf16_t)(r0 & 0xffff)), hv_bf2f((bf16_t)(r0 >> 16)), hv_bf2f((bf16_t)(r1 & 0xffff)),
                                   hv_bf2f((bf16_t)(r1 >> 16))};
                    }
                    o[2 * k] = hv_pack2(v[0], v[1]);
                    o[2 * k + 1] = hv_pack2(v[2], v[3]);
                    if (STATS == 1 && m_base + 16 * mf + r16 < p.M) {
                        gs[nf] += v;
                        gq[nf] += v * v;
                    }
                    if (STATS == 2 && nc[h] == n_base + 32 * h + 8 * quad) {  // (not a clamped duplicate of the ragged edge)
                        rs[mf] += (v[0] + v[1]) + (v[2] + v[3]);
                        rq[mf] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                    }
                }
                outp[mf][h] = o;
            }
        }
#if !defined(HV_EMU)
        __builtin_amdgcn_sched_barrier(0);  // ... and the next group's loads are not hoisted over it (register budget)
#endif
    }
#ifndef HV_EMU
    // (pinned results + "no load outstanding": see hv_gemm_epilogue_fast)
#pragma unroll
    for (int mf = 0; mf < NMF; ++mf)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            asm volatile("" : "+v"(outp[mf][h][0]), "+v"(outp[mf][h][1]), "+v"(outp[mf][h][2]), "+v"(outp[mf][h][3]));
    __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
    char* const yb = reinterpret_cast<char*>(p.Y);
#pragma unroll
    for (int mf = 0; mf < NMF; ++mf) {
        const int m = m_base + 16 * mf + r16;
        if (m >= p.M) continue;
        const unsigned yo = (unsigned)mo[mf] * (unsigned)p.ldy * 2u;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = n_base + 32 * h + 8 * quad;
            if (n >= p.N) continue;
            hv_st16(yb + (yo + 2u * (unsigned)n), outp[mf][h]);
        }
    }
    if (STATS == 2 && p.ln_part != nullptr && n_base < p.N) {  // (wave-uniform; a wave beyond the ragged N edge has no block)
        const int parts = p.N / 64, blk = n_base / 64;
#pragma unroll
        for (int mf = 0; mf < NMF; ++mf) {
            float a = rs[mf], b = rq[mf];
            a += __shfl_xor(a, 16);
            b += __shfl_xor(b, 16);
            a += __shfl_xor(a, 32);
            b += __shfl_xor(b, 32);
            const int m = m_base + 16 * mf + r16;
            if (quad == 0 && m < p.M) *reinterpret_cast<u32x2*>(p.ln_part + ((long)mo[mf] * parts + blk) * 2) =
                u32x2{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
        }
    }
    if (STATS == 1 && p.gn_part != nullptr && m_base < p.M) {  // (wave-uniform) rows [m_base, + 16 NMF) lie in one image: hv_gemm_gn_parts;
        // a wave sub-tile that starts beyond the ragged M edge (M % 128 == 64 on the 128 x 128 kernel) has no part to write
        const int rows = 16 * NMF, parts = p.gn_rows_per_image / rows;
        const int img = m_base / p.gn_rows_per_image, part = (m_base - img * p.gn_rows_per_image) / rows;
        float* dst = p.gn_part + ((long)img * parts + part) * p.N * 2;
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            f32x4 a, b;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] = hv_row16_sum(gs[nf][e]);
                b[e] = hv_row16_sum(gq[nf][e]);
            }
            const int n = n_base + 32 * (nf >> 1) + 8 * quad + 4 * (nf & 1);
            if (r16 == 0 && n < p.N) {
                *reinterpret_cast<f32x4*>(dst + 2 * n) = f32x4{a[0], b[0], a[1], b[1]};
                *reinterpret_cast<f32x4*>(dst + 2 * n + 4) = f32x4{a[2], b[2], a[3], b[3]};
            }
        }
    }
}

// LayerNorm fold + GEGLU under hv_perm_row_geglu: fragments (0, 1) = h, g of output channels n_base / 2 + 8 quad + 0..3,
// fragments (2, 3) = h, g of the next four.  N % 32 == 0 (checked by hv_gemm_launch for every GEGLU problem).
template <int NMF>
HV_DEV void hv_gemm_epilogue_fast_perm_geglu(const HvGemmParams& p, f32x4 (&acc)[4][NMF], int m_base, int n_base, int r16, int quad,
                                             const float* tab_row) {
    constexpr int G = NMF < HV_GEMM_EPI_G ? NMF : HV_GEMM_EPI_G;
    f32x4 add4[4], cs4[4];
    auto ld4 = [&](const float* base, unsigned byte_ofs) __attribute__((always_inline)) {
        return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + byte_ofs);
    };
    unsigned nb[4];  // packed column of the fragment's first row of this lane (4 consecutive packed columns follow)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) nb[nf] = 4u * (unsigned)min(n_base + hv_perm_row_geglu(nf, 4 * quad), p.N - 4);
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
        add4[nf] = p.bias != nullptr ? ld4(p.bias, nb[nf]) : f32x4{0.f, 0.f, 0.f, 0.f};
        cs4[nf] = ld4(p.colsum, nb[nf]);
    }
    if (tab_row != nullptr) {
        f32x4 t4[4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) t4[nf] = ld4(tab_row, nb[nf]);
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) add4[nf] += t4[nf];
    }
    u32x4 outp[NMF];
#pragma unroll
    for (int g = 0; g < NMF; g += G) {
        float mean[G], rstd[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int mc = min(m_base + 16 * (g + j) + r16, p.M - 1);
            mean[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.row_mean) + 4u * (unsigned)mc);
            rstd[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.row_rstd) + 4u * (unsigned)mc);
        }
#if !defined(HV_EMU)
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int mf = g + j;
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x4 h = rstd[j] * (acc[2 * k][mf] - mean[j] * cs4[2 * k]) + add4[2 * k];
                const f32x4 gt = rstd[j] * (acc[2 * k + 1][mf] - mean[j] * cs4[2 * k + 1]) + add4[2 * k + 1];
                h = hv_gelu_times(gt, h);
                o[2 * k] = hv_pack2(h[0], h[1]);
                o[2 * k + 1] = hv_pack2(h[2], h[3]);
            }
            outp[mf] = o;
        }
#if !defined(HV_EMU)
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
#ifndef HV_EMU
#pragma unroll
    for (int mf = 0; mf < NMF; ++mf) asm volatile("" : "+v"(outp[mf][0]), "+v"(outp[mf][1]), "+v"(outp[mf][2]), "+v"(outp[mf][3]));
    __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
    char* const yb = reinterpret_cast<char*>(p.Y);
    const int no = (n_base >> 1) + 8 * quad;  // output channel of the lane's first result
#pragma unroll
    for (int mf = 0; mf < NMF; ++mf) {
        const int m = m_base + 16 * mf + r16;
        if (m >= p.M || 2 * no >= p.N) continue;
        hv_st16(yb + ((unsigned)m * (unsigned)p.ldy * 2u + 2u * (unsigned)no), outp[mf]);
    }
}

template <int NMF>
HV_DEV void hv_gemm_epilogue(const HvGemmParams& p, f32x4 (&acc)[4][NMF], int m_base, int n_base, int r16, int quad) {
    const bool lean = p.out_act == HV_ACT_NONE && !p.out_f32 && p.perm_p == 0;
    if (lean && p.geglu)
        hv_gemm_epilogue_t<NMF, 3>(p, acc, m_base, n_base, r16, quad);
    else if (lean && p.Yt != nullptr)
        hv_gemm_epilogue_t<NMF, 2>(p, acc, m_base, n_base, r16, quad);
    else if (lean)
        hv_gemm_epilogue_t<NMF, 1>(p, acc, m_base, n_base, r16, quad);
    else
        hv_gemm_epilogue_t<NMF, 0>(p, acc, m_base, n_base, r16, quad);
}

// Output forms of hv_gemm_epilogue_fast that the LDS-DMA kernel instantiates; hv_gemm_fast_form() (host) classifies a
// problem, everything else runs on the register-staged kernel with the general epilogue.
enum { HV_FORM_NONE = -1, HV_FORM_LN = 0, HV_FORM_LN_YT = 1, HV_FORM_LN_GEGLU = 2, HV_FORM_RES = 3, HV_FORM_PLAIN = 4 };

static inline int hv_gemm_fast_form(const HvGemmParams& p, int rows_per_wave) {
    if (p.out_act != HV_ACT_NONE || p.out_f32) return HV_FORM_NONE;
    // a row permutation of the output: only the permuted-channel epilogue of the plain-output forms knows it
    // (hv_gemm_epilogue_fast_perm; hv_gemm_choose checks that this epilogue is the one that runs), in steps of 16 rows
    if (p.perm_p != 0 && (p.perm_p < 16 || p.geglu || p.Yt != nullptr || p.gn_part != nullptr)) return HV_FORM_NONE;
    if (p.pe != nullptr && p.rowvec != nullptr) return HV_FORM_NONE;
    // one table row per wave sub-tile: sub-tiles start at multiples of rows_per_wave
    if (p.pe != nullptr && (p.pe_period <= 0 || p.pe_period % rows_per_wave != 0)) return HV_FORM_NONE;
    if (p.rowvec != nullptr && (p.rowvec_period <= 0 || p.rowvec_period % rows_per_wave != 0)) return HV_FORM_NONE;
    const bool ln = p.row_rstd != nullptr, res = p.residual != nullptr;
    if (p.geglu) return (ln && !res) ? HV_FORM_LN_GEGLU : HV_FORM_NONE;
    if (p.Yt != nullptr) return (ln && !res) ? HV_FORM_LN_YT : HV_FORM_NONE;
    if (ln) return res ? HV_FORM_NONE : HV_FORM_LN;  // LayerNorm fold + residual: not on the denoising path
    return res ? HV_FORM_RES : HV_FORM_PLAIN;
}

template <int NMF, bool PERM = false, int STATS = 0>
HV_DEV void hv_gemm_epilogue_form(int form, const HvGemmParams& p, f32x4 (&acc)[4][NMF], int m_base, int n_base, int r16,
                                  int quad) {
#ifndef HV_EMU
    const int m_first = __builtin_amdgcn_readfirstlane(min(m_base, p.M - 1));  // wave-uniform: the table row is a scalar base
#else
    const int m_first = min(m_base, p.M - 1);
#endif
    const float* tab = nullptr;
    if (p.pe != nullptr) tab = p.pe + (long)((m_first / p.pe_period) % p.pe_frames) * p.N;
    else if (p.rowvec != nullptr) tab = p.rowvec + (long)(m_first / p.rowvec_period) * p.N;
    if constexpr (PERM) {  // the launcher sends only the plain-output forms here
        switch (form) {
            case HV_FORM_LN_GEGLU: hv_gemm_epilogue_fast_perm_geglu<NMF>(p, acc, m_base, n_base, r16, quad, tab); break;
            case HV_FORM_LN: hv_gemm_epilogue_fast_perm<NMF, true, false>(p, acc, m_base, n_base, r16, quad, tab); break;
            case HV_FORM_RES: hv_gemm_epilogue_fast_perm<NMF, false, true, STATS>(p, acc, m_base, n_base, r16, quad, tab); break;
            default: hv_gemm_epilogue_fast_perm<NMF, false, false, STATS>(p, acc, m_base, n_base, r16, quad, tab); break;
        }
        return;
    }
    switch (form) {
        case HV_FORM_LN: hv_gemm_epilogue_fast<NMF, true, false, 0>(p, acc, m_base, n_base, r16, quad, tab); break;
        case HV_FORM_LN_YT: hv_gemm_epilogue_fast<NMF, true, false, 1>(p, acc, m_base, n_base, r16, quad, tab); break;
        case HV_FORM_LN_GEGLU: hv_gemm_epilogue_fast<NMF, true, false, 2>(p, acc, m_base, n_base, r16, quad, tab); break;
        case HV_FORM_RES: hv_gemm_epilogue_fast<NMF, false, true, 0>(p, acc, m_base, n_base, r16, quad, tab); break;
        default: hv_gemm_epilogue_fast<NMF, false, false, 0>(p, acc, m_base, n_base, r16, quad, tab); break;
    }
}

template <int Q>
HV_DEV unsigned& hv_pick4(unsigned& a, unsigned& b, unsigned& c, unsigned& d) {
    if constexpr (Q == 0) return a;
    else if constexpr (Q == 1) return b;
    else if constexpr (Q == 2) return c;
    else return d;
}

template <int N>
struct HvInt {
    static constexpr int value = N;
};

// (Round 4, measured and removed: starting the workgroups of an XCD a fraction of a tile apart -- 2 / 4 / 8 phase groups -- so that
//  the CUs' epilogue store bursts do not coincide.  Same-box A/B, profiles/r04_s2.txt: every GEMM shape of the step within
//  -3 % .. +12 % of the unstaggered time, the step 109.4 -> 110.2 / 110.5 / 111.9 ms: the "store issue" share of a tile
//  (profiles/r03_gemm_trace.txt) is not a burst effect.)
// Persistent workgroups: each walks a strided list of output tiles of its XCD's contiguous tile
// range; the (tile, k-step) sequence is flattened so that the register prefetch (two k-tiles in
// flight per workgroup) runs across tile boundaries and the epilogue of tile i overlaps the loads
// of tile i+1.  The GEMMs of this path have short K (320..1280, 5..20 k-steps): without this the
// kernel is L2-latency bound (one 32 KiB k-tile in flight per workgroup).
template <int BK>
__global__ __launch_bounds__(256, 2) void hv_gemm_kernel(HvGemmParams p) {
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

    // XCD-aware persistent tile walk: XCD x owns tiles [x*per_xcd, (x+1)*per_xcd); its workgroups
    // take consecutive tiles (all N-tiles of an M-panel run together and share the panel in L2)
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int total = tiles_n * tiles_m;
    const int wg_per_xcd = gridDim.x / 8;
    const int xcd = blockIdx.x % 8, wg = blockIdx.x / 8;
    const int per_xcd = (total + 7) / 8;
    const int t_begin = xcd * per_xcd;
    const int t_end = min(total, t_begin + per_xcd);
    const int first = t_begin + wg;
    if (first >= t_end) return;
    const int my_tiles = (t_end - first + wg_per_xcd - 1) / wg_per_xcd;
    const int nk = p.K / BK;
    const int nsteps = my_tiles * nk;

    u32x4 xr[2][CH_PER_THREAD], wr[2][CH_PER_THREAD];

    auto load_step = [&](int s, auto SET) __attribute__((always_inline)) {
        constexpr int R = decltype(SET)::value;
        const int ti = first + (s / nk) * wg_per_xcd;
        const int m0 = (ti / tiles_n) * BM, n0 = (ti % tiles_n) * BN;
        const int k0 = (s % nk) * BK;
#pragma unroll
        for (int i = 0; i < CH_PER_THREAD; ++i) {
            const int id = tid + 256 * i;
            const int row = id / CPR, c = id % CPR;
            const int m = m0 + row, n = n0 + row, k = k0 + c * 8;
            u32x4 z = {0u, 0u, 0u, 0u};
            xr[R][i] = z;
            wr[R][i] = z;
            if (m < p.M) {
                const bf16_t* src = (p.X2 != nullptr && k >= p.K1) ? p.X2 + (long)m * p.ldx2 + (k - p.K1)
                                                                   : p.X + (long)m * p.ldx + k;
                xr[R][i] = hv_ld16(src);
            }
            if (n < p.N) wr[R][i] = hv_ld16(p.W + (long)n * p.K + k);
        }
    };

    auto store_step = [&](int s, auto SET) __attribute__((always_inline)) {
        constexpr int R = decltype(SET)::value;  // register set == LDS buffer (s & 1)
        const int ti = first + (s / nk) * wg_per_xcd;
        const int m0 = (ti / tiles_n) * BM;
        const int k0 = (s % nk) * BK;
#pragma unroll
        for (int i = 0; i < CH_PER_THREAD; ++i) {
            const int id = tid + 256 * i;
            const int row = id / CPR, c = id % CPR;
            u32x4 xv = xr[R][i];
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
            hv_st16(Xs + R * TILE_BYTES + hv_swz<BK>(row, c), xv);
            hv_st16(Ws + R * TILE_BYTES + hv_swz<BK>(row, c), wr[R][i]);
        }
    };

    f32x4 acc[4][4];  // [nf][mf]
    auto clear_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    auto compute = [&](auto SET) __attribute__((always_inline)) {
        constexpr int R = decltype(SET)::value;
        const unsigned char* xs = Xs + R * TILE_BYTES;
        const unsigned char* ws = Ws + R * TILE_BYTES;
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
    };

    auto epilogue = [&](int ti) __attribute__((always_inline)) {
        hv_gemm_epilogue<4>(p, acc, (ti / tiles_n) * BM + 64 * wm, (ti % tiles_n) * BN + 64 * wn, r16, quad);
    };

    // one flattened step: park k-tile s in LDS, refill its registers with k-tile s+2, multiply,
    // and finish the tile when its last k-tile has been consumed
    auto step = [&](int s, auto SET) __attribute__((always_inline)) {
        store_step(s, SET);
        __syncthreads();
        if (s + 2 < nsteps) load_step(s + 2, SET);
        compute(SET);
        if ((s + 1) % nk == 0) {
            epilogue(first + (s / nk) * wg_per_xcd);
            clear_acc();
        }
    };

    clear_acc();
    load_step(0, HvInt<0>());
    if (nsteps > 1) load_step(1, HvInt<1>());
    for (int s = 0; s < nsteps; s += 2) {
        step(s, HvInt<0>());
        if (s + 1 < nsteps) step(s + 1, HvInt<1>());
    }
}

// ---- LDS-DMA kernel (no operand prologue): BM x BN x 64 tiles on a 2-slot LDS ring filled with global_load_lds (no VGPR
// staging, no ds_write: the register-staged kernel above is bound by the LDS write path), persistent tile walk as above,
// one raw s_barrier per half k-step, counted vmcnt waits -- never vmcnt(0) inside the loop.  Two instantiations:
//   256 x 256 x 64, 8 waves (2 x 4 of 128 x 64), 128 KiB ring, one workgroup per CU: a third fewer operand bytes per FLOP
//       through the per-CU L2 -> LDS fill path, which is what bounds this kernel (profiles/r02_gemm_trace.txt);
//   128 x 128 x 64, 4 waves (2 x 2 of 64 x 64), 64 KiB ring, two workgroups per CU: finer quantisation over the 256 CUs
//       for narrow / short problems, one workgroup's epilogue under the other's k-loop.
// The k-tile's LDS-DMA instructions go out as two readiness groups (cdna_hip_programming.md T3+T4) -- see the k-loop.
//   PH = 1: each group as a burst at the head of its half step (measured best for the 256 x 256 tile, round 3),
//   PH = 2: one DMA instruction in front of every 8 MFMAs (measured best for the 128 x 128 tile).
// Both are bit-identical to the round-2 one-burst loop (same MFMA order per accumulator; hardware-checked in
// profiles/r03_hwcheck.txt), which is deleted together with the BK = 32 / 3-slot / contiguous-walk / L2-prefetch variants
// that never won a same-box A/B.
//   PERM: the permuted channel assignment of a wave's 64-channel block (hv_perm_row) for the plain-output forms.
//   STATS (with PERM): the plain / residual epilogue also leaves GroupNorm (1, p.gn_part) or LayerNorm (2, p.ln_part) partial
//   statistics of its tile.
template <int BN, int NW, int BM, int PH, bool PERM = false, int STATS = 0>
__global__ __launch_bounds__(NW * 64, 2) void hv_gemm_glds_kernel(HvGemmParams p, int gm, int form) {
    constexpr int BK = 64, NS = 2;
    constexpr int WAVES_N = BN / 64, WAVES_M = NW / WAVES_N;
    constexpr int WTM = BM / WAVES_M, NMF = WTM / 16;
    constexpr int XT = BM * BK * 2, WT = BN * BK * 2, SLOT = XT + WT;
    constexpr int RPI = 1024 / (BK * 2);        // tile rows covered by one 1 KiB wave-instruction
    constexpr int CPR = BK / 8, RPB = 16 / CPR;  // 16-byte chunks per row, rows per 256-byte bank row
    constexpr int XQ = BM / RPI / NW, WQ = BN / RPI / NW;  // DMA instructions per wave and k-tile
    static_assert(PH == 1 || PH == 2, "issue cadence");
    static_assert(XQ == 4 && WQ == 4 && WAVES_M == 2 && NMF % 2 == 0,
                  "the two-group k-loop is written for the 256 x 256 x 64 tile on 8 waves and the 128 x 128 x 64 tile on 4");
    __shared__ __attribute__((aligned(16))) unsigned char smem[NS * SLOT];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifndef HV_EMU
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the LDS-DMA destinations are wave-uniform
#else
    const int wave = tid >> 6;
#endif
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int r16 = lane & 15, quad = lane >> 4;

    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int total = tiles_n * tiles_m;
    // Tile raster.  The workgroups of an XCD walk consecutive tile indices at the same time and share its 4 MiB L2.
    // Row-major tile order makes them n-tiles of ONE m-block when N is wide: every m-block then re-streams the whole
    // weight matrix through the L2-miss path (measured with FETCH_SIZE: 27x the algorithmic read bytes at N = 10240).
    // Walking gm m-blocks per n-step instead makes the concurrent set gm x (64/gm) tiles: each X k-slice is shared by
    // 64/gm workgroups and each W k-slice by gm.
    auto tile_origin = [&](int ti, int& m0, int& n0) __attribute__((always_inline)) {
        if (gm <= 1) {
            m0 = (ti / tiles_n) * BM;
            n0 = (ti % tiles_n) * BN;
            return;
        }
        const int per_group = gm * tiles_n;
        const int g = ti / per_group, r = ti - g * per_group;
        const int rows = max(1, min(gm, tiles_m - g * gm));  // the last group may be shorter (ti may run past the end: unused)
        m0 = (g * gm + r % rows) * BM;
        n0 = (r / rows) * BN;
    };
    // workgroup w of an XCD takes tiles w, w + G, w + 2G ... of the XCD's contiguous range
    const int wg_per_xcd = gridDim.x / 8;
    const int xcd = blockIdx.x % 8, wg = blockIdx.x / 8;
    const int per_xcd = (total + 7) / 8;
    const int t_begin = xcd * per_xcd;
    const int t_end = min(total, t_begin + per_xcd);
    const int first = t_begin + wg;
    const int tstep = wg_per_xcd;
    if (first >= t_end) return;
    const int my_tiles = (t_end - first + wg_per_xcd - 1) / wg_per_xcd;
    const int nk = p.K / BK;
    const int nsteps = my_tiles * nk;

    // LDS-DMA issue state, advanced one k-tile per call.  The per-lane source address of each of the wave's DMA
    // instructions is computed once per tile and advanced by BK elements per k-step; the k-loop needs no kernel
    // argument besides them.  The row -> bank-row swizzle only depends on (row / RPB) % CPR.
    auto chunk_ofs_s = [&](int j, int sub) __attribute__((always_inline)) {
        const int row = (RPI % 16 == 0) ? sub : RPI * j + sub;
        return (((lane & (CPR - 1))) ^ ((row / RPB) % CPR)) * 8;
    };
    const int k1_steps = p.X2 != nullptr ? p.K1 / BK : -1;  // k-tile at which the second source takes over
    int i_tile = first, i_k = 0, i_slot = 0;
    int i_m0, i_n0;
    // 32-bit byte offsets from a wave-uniform base (hv_gemm_launch checks that the operands span < 4 GiB): half the
    // registers of full pointers, and the DMA instruction takes the base from an SGPR pair
    unsigned xo0 = 0, xo1 = 0, xo2 = 0, xo3 = 0, wo0 = 0, wo1 = 0, wo2 = 0, wo3 = 0;  // named scalars: see the note below
    const char* xbase = reinterpret_cast<const char*>(p.X);
    const char* const wbase = reinterpret_cast<const char*>(p.W);
    // (named scalars picked by a compile-time index: as arrays hipcc kept the offsets in a stack slot: a scratch reload
    //  behind vmcnt(0) in every k-step)
    // (tile changes are rare: keep their lane constants OUT of the k-loop's register budget -- an opaque copy per call stops
    //  hipcc from hoisting "RPI * j + sub" for every j into loop-carried registers, which spilled in the 256 x 256 kernel)
    auto fresh = [&](int v) __attribute__((always_inline)) {
#ifndef HV_EMU
        asm volatile("" : "+v"(v));
#endif
        return v;
    };
    auto set_x = [&](const bf16_t* base, long ld) __attribute__((always_inline)) {
        const int sub = fresh(lane / CPR);
        xbase = reinterpret_cast<const char*>(base);
        hv_static_for<XQ>([&](auto Q) __attribute__((always_inline)) {
            constexpr int q = decltype(Q)::value;
            const int j = wave + NW * q;
            const int m = min(i_m0 + RPI * j + sub, p.M - 1);
            hv_pick4<q>(xo0, xo1, xo2, xo3) = ((unsigned)m * (unsigned)ld + (unsigned)chunk_ofs_s(j, sub)) * 2u;  // < 4 GiB: exact in 32 bits
        });
    };
    auto set_tile = [&]() __attribute__((always_inline)) {
        tile_origin(i_tile, i_m0, i_n0);
        if (k1_steps == 0) set_x(p.X2, p.ldx2);
        else set_x(p.X, p.ldx);
        const int sub = fresh(lane / CPR);
        hv_static_for<WQ>([&](auto Q) __attribute__((always_inline)) {
            constexpr int q = decltype(Q)::value;
            const int j = wave + NW * q;
            const int n = min(i_n0 + RPI * j + sub, p.N - 1);
            // source-side swizzle of the lane's 16-byte chunk: LDS chunk slot (lane % CPR) of tile row RPI j + sub holds
            // the chunk whose index is slot ^ swizzle(row) -- the W tile's own swizzle under PERM
            const int trow = RPI * j + sub;
            const int wsw = PERM ? hv_wperm_swizzle(trow) : ((trow / RPB) % CPR);
            hv_pick4<q>(wo0, wo1, wo2, wo3) = ((unsigned)n * (unsigned)p.K + (unsigned)(((lane & (CPR - 1)) ^ wsw) * 8)) * 2u;
        });
    };
    set_tile();
    // one DMA instruction of the k-tile being issued (X rows BM/4 q .. +BM/4 / W rows likewise, over the NW waves) ...
    auto issue_x1 = [&](auto Q) __attribute__((always_inline)) {
        constexpr int q = decltype(Q)::value;
        unsigned& o = hv_pick4<q>(xo0, xo1, xo2, xo3);
        hv_glds16(xbase + o, smem + i_slot * SLOT + (wave + NW * q) * 1024);
        o += BK * 2;
    };
    auto issue_w1 = [&](auto Q) __attribute__((always_inline)) {
        constexpr int q = decltype(Q)::value;
        unsigned& o = hv_pick4<q>(wo0, wo1, wo2, wo3);
        hv_glds16(wbase + o, smem + i_slot * SLOT + XT + (wave + NW * q) * 1024);
        o += BK * 2;
    };
    // ... and the step to the next k-tile once all XQ + WQ of them are out
    auto issue_advance = [&]() __attribute__((always_inline)) {
        if (++i_slot == NS) i_slot = 0;
        if (++i_k == nk) {
            i_k = 0;
            i_tile += tstep;
            set_tile();
        } else if (i_k == k1_steps) {
            set_x(p.X2, p.ldx2);
        }
    };

    f32x4 acc[4][NMF];  // [nf][mf]
    auto clear_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < NMF; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    clear_acc();
    {  // prologue: k-tile 0 in readiness-group order (G0 = all of W + the X rows of the first fragment half, G1 = the rest)
        hv_static_for<WQ>([&](auto Q) __attribute__((always_inline)) { issue_w1(Q); });
        issue_x1(HvInt<0>{});
        issue_x1(HvInt<2>{});
        issue_x1(HvInt<1>{});
        issue_x1(HvInt<3>{});
        issue_advance();
    }
    int c_tile = first, c_k = 0, c_slot = 0;  // consumer state
    int landed = 0;  // k-steps that need no vmcnt wait (see below)
    constexpr int HMF = NMF / 2;  // fragment rows per half: X DMA instruction q covers rows [BM / 4 * q, +BM / 4) = half q % 2 of wm = q / 2
    for (int s = 0; s < nsteps; ++s) {
        // Two readiness groups per k-tile, counted vmcnt, no drain.  A wave multiplies the X rows [WTM wm, +WTM) with the
        // W rows [64 wn, +64).  DMA instruction q of the NW waves covers rows [BM / 4 q, +BM / 4) of its operand, so
        //   G0 = W q=0..3, X q=0, X q=2 : everything the FIRST fragment half (X rows WTM wm + 0 .. WTM/2 - 1) needs,
        //   G1 = X q=1, X q=3           : the X rows of the second half.
        // Each wave issues the k-tile in that order (vmcnt retires in order): G0 of k-tile s+1 during the first half of step
        // s, G1 of it during the second.  Barrier B0 (step boundary) needs G0(s): the two G1(s) instructions behind it stay
        // in flight -> vmcnt(2); barrier B1 (mid-step) needs G1(s): the six G0(s+1) instructions behind it stay in flight
        // -> vmcnt(6) (PH = 2: only the four W instructions are out by then -> vmcnt(4)).  Slot (s+1) % 2 was last read in
        // step s-1, i.e. before B0(s): free for the whole of step s.
        // After an epilogue everything has landed (it waits for every load before its first store): both waits of the next
        // step are skipped, and the stores drain under it (they are older than that step's DMA, so the counted waits of the
        // step after it retire them first -- a full k-step later).
        const bool more = s + 1 < nsteps;
        const bool skip = landed > 0;
        if (skip) --landed;
        else hv_vm_wait<2>();
        hv_barrier_raw();
        const unsigned char* xs = smem + c_slot * SLOT;
        const unsigned char* ws = xs + XT;
        if (++c_slot == NS) c_slot = 0;
        if (PH == 1 && more) {
            issue_w1(HvInt<0>{});
            issue_w1(HvInt<1>{});
            issue_w1(HvInt<2>{});
        }
        bf16x8 wf[2][4];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                // PERM: the lane's W-tile row per fragment depends on the output form (wave-uniform, loop-invariant)
                const int wrow = !PERM ? 16 * f + r16 : (form == HV_FORM_LN_GEGLU ? hv_perm_row_geglu(f, r16) : hv_perm_row(f, r16));
                wf[kk][f] = hv_as_bf16x8(hv_ld16(ws + (PERM ? hv_swz_wperm(64 * wn + wrow, kk * 4 + quad)
                                                            : hv_swz<BK>(64 * wn + wrow, kk * 4 + quad))));
            }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 xf[HMF];
#pragma unroll
            for (int f = 0; f < HMF; ++f) xf[f] = hv_as_bf16x8(hv_ld16(xs + hv_swz<BK>(WTM * wm + 16 * f + r16, kk * 4 + quad)));
            if (PH == 1 && kk == 1 && more) {
                issue_w1(HvInt<3>{});
                issue_x1(HvInt<0>{});
                issue_x1(HvInt<2>{});
            }
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                if (PH == 2 && more && (nf & 1) == 0) {
                    if (kk == 0 && nf == 0) issue_w1(HvInt<0>{});
                    if (kk == 0 && nf == 2) issue_w1(HvInt<1>{});
                    if (kk == 1 && nf == 0) issue_w1(HvInt<2>{});
                    if (kk == 1 && nf == 2) issue_w1(HvInt<3>{});
                }
#pragma unroll
                for (int mf = 0; mf < HMF; ++mf)
                    acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][nf], xf[mf], acc[nf][mf], 0, 0, 0);
            }
        }
        if (!skip) {
            if (!more) hv_vm_wait<0>();
            else if (PH == 2) hv_vm_wait<4>();
            else hv_vm_wait<6>();
        }
        hv_barrier_raw();
        if (PH == 1 && more) issue_x1(HvInt<1>{});
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 xf[HMF];
#pragma unroll
            for (int f = 0; f < HMF; ++f)
                xf[f] = hv_as_bf16x8(hv_ld16(xs + hv_swz<BK>(WTM * wm + 16 * HMF + 16 * f + r16, kk * 4 + quad)));
            if (PH == 1 && kk == 1 && more) {
                issue_x1(HvInt<3>{});
                issue_advance();
            }
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                if (PH == 2 && more && (nf & 1) == 0) {
                    if (kk == 0 && nf == 0) issue_x1(HvInt<0>{});
                    if (kk == 0 && nf == 2) issue_x1(HvInt<2>{});
                    if (kk == 1 && nf == 0) issue_x1(HvInt<1>{});
                    if (kk == 1 && nf == 2) {
                        issue_x1(HvInt<3>{});
                        issue_advance();
                    }
                }
#pragma unroll
                for (int mf = 0; mf < HMF; ++mf)
                    acc[nf][HMF + mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][nf], xf[mf], acc[nf][HMF + mf], 0, 0, 0);
            }
        }
        if (++c_k == nk) {
            c_k = 0;
            {
                int m0, n0;
                tile_origin(c_tile, m0, n0);
                hv_gemm_epilogue_form<NMF, PERM, STATS>(form, p, acc, m0 + WTM * wm, n0 + 64 * wn, r16, quad);
                landed = 1;
            }
            c_tile += tstep;
            clear_acc();
        }
    }
}

// ---- epilogue of the wide-tile kernel (below): one wave's 32 rows x 320 columns = five 64-column blocks under the permuted
// channel assignment (hv_perm_row: the fragment pair (2 j, 2 j + 1) of a lane is the 8 consecutive channels 64 b + 32 j + 8 quad).
// Round 3 called hv_gemm_epilogue_fast_perm once per block: the five copies of the run-time form switch and of the pointer
// tests became branches around every block with accumulators spilled across them (19-30 registers), and every block's
// residual request waited for its own HBM round trip.  Here the forms are compile-time, there is no pointer test inside
// (the bias is mandatory for this kernel; a problem without a per-row table passes the bias row again with weight 0), the
// residual rows of block b + 1 are requested BEFORE block b is converted (two register sets alternating at compile time)
// and a block's results are stored as soon as they are packed (160 accumulator registers leave no room to hold them back).
// No statistics variants: no N = 320 / K >= 640 output of the denoising path feeds a norm (the parts queries answer 0 for this
// kernel, so a caller that wants statistics gets the statistics pass).
template <bool LN, bool RES>
HV_DEV void hv_gemm_epilogue_wide(const HvGemmParams& p, f32x4 (&acc)[5][4][2], int m_base, int n0, int r16, int quad,
                                  const float* tab_row, float tab_scale) {
    constexpr int NB = 5, NMF = 2;
    static_assert(!(LN && RES), "forms of hv_gemm_fast_form");
    auto ld4 = [&](const float* base, unsigned byte_ofs) __attribute__((always_inline)) {
        return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + byte_ofs);
    };
#ifndef HV_EMU
    // the LDS-DMA of the next k-tile (inline asm: invisible to hipcc's wait counts) has landed before anything below is
    // issued: the caller skips its own wait after an epilogue
    __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
    float mean[NMF], rstd[NMF];
    unsigned rrow[NMF], yrow[NMF];  // byte offsets of the lane's residual / output rows
#pragma unroll
    for (int mf = 0; mf < NMF; ++mf) {
        const unsigned m = (unsigned)(m_base + 16 * mf + r16);  // M % 256 == 0: no ragged rows
        mean[mf] = rstd[mf] = 0.f;
        if (LN) {
            mean[mf] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.row_mean) + 4u * m);
            rstd[mf] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.row_rstd) + 4u * m);
        }
        rrow[mf] = m * (unsigned)p.ldr * 2u;
        yrow[mf] = m * (unsigned)p.ldy * 2u;
    }
    u32x4 resA[NMF][2], resB[NMF][2];
    auto load_res = [&](int b, u32x4(&r)[NMF][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int mf = 0; mf < NMF; ++mf)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                r[mf][h] = hv_ld16(reinterpret_cast<const char*>(p.residual) + (rrow[mf] + 2u * (unsigned)(n0 + 64 * b + 32 * h + 8 * quad)));
    };
    char* const yb = reinterpret_cast<char*>(p.Y);
    if (RES) load_res(0, resA);
    hv_static_for<NB>([&](auto B) __attribute__((always_inline)) {
        constexpr int b = decltype(B)::value;
        u32x4(&cur)[NMF][2] = (b % 2 == 0) ? resA : resB;
        u32x4(&nxt)[NMF][2] = (b % 2 == 0) ? resB : resA;
        if constexpr (RES && b + 1 < NB) load_res(b + 1, nxt);
#if !defined(HV_EMU)
        __builtin_amdgcn_sched_barrier(0);  // the next block's residual requests stay ahead of this block's arithmetic
#endif
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4 o[NMF];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                // per-column vectors: channels 64 b + 32 h + 8 quad + 4 k .. + 3 (L1 / L2 hits)
                const unsigned co = 4u * (unsigned)(n0 + 64 * b + 32 * h + 8 * quad + 4 * k);
                const f32x4 add = ld4(p.bias, co) + tab_scale * ld4(tab_row, co);
                const f32x4 cs = LN ? ld4(p.colsum, co) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mf = 0; mf < NMF; ++mf) {
                    f32x4 v = acc[b][2 * h + k][mf];
                    if (LN) v = rstd[mf] * (v - mean[mf] * cs);
                    v += add;
                    if (RES) {
                        const unsigned r0 = cur[mf][h][2 * k], r1 = cur[mf][h][2 * k + 1];
                        v += f32x4{hv_bf2f((bf16_t)(r0 & 0xffff)), hv_bf2f((bf16_t)(r0 >> 16)), hv_bf2f((bf16_t)(r1 & 0xffff)),
                                   hv_bf2f((bf16_t)(r1 >> 16))};
                    }
                    const unsigned o0 = hv_pack2(v[0], v[1]), o1 = hv_pack2(v[2], v[3]);
                    o[mf][2 * k] = o0;
                    o[mf][2 * k + 1] = o1;
                }
            }
#pragma unroll
            for (int mf = 0; mf < NMF; ++mf) hv_st16(yb + (yrow[mf] + 2u * (unsigned)(n0 + 64 * b + 32 * h + 8 * quad)), o[mf]);
        }
#if !defined(HV_EMU)
        __builtin_amdgcn_sched_barrier(0);
#endif
    });
}

// ---- wide-tile LDS-DMA kernel for narrow outputs (round 3, written at the end of the round without GPU time to tune it:
// OPT-IN, hv_set_tuning(HV_TUNE_GEMM_GLDS, 4)): 256 x 320 x 64 tiles for N = 320 (the kernel itself handles any N % 320 == 0).  The 128 x 128 kernel spends three
// column tiles on 320 columns (17 % of its MFMAs and W bytes on padding) and streams the X rows once per column tile through
// the CU's L2 -> LDS fill path, which is what bounds it (profiles/r03_gemm_trace.txt): here a 256-row block of X is read
// ONCE and multiplied with all 320 columns -- 72 KiB of fill per k-step for 256 x 320 x 64 MACs against 192 KiB.
//   8 waves; wave w owns rows [32 w, +32) x all 320 columns = 5 column blocks of 64 (the epilogues' unit): 2 x 20 fragments,
//   160 accumulator registers.  Ring: 2 slots x (X 32 KiB + W 40 KiB) = 144 KiB, one workgroup per CU.
//   One raw barrier per k-step: wait for the own DMA of k-tile s, barrier (everyone's k-tile s landed; everyone is done with
//   slot (s+1) % 2), issue k-tile s+1, multiply from slot s % 2.  The DMA goes out from inline asm (hv_glds16_s): hipcc
//   does not see it and puts no vmcnt(0) in front of the fragment reads.
//   Same swizzles, permuted channel assignment and epilogues (per 64-column block, NMF = 2) as hv_gemm_glds_kernel.
template <int BN>  // (a template also keeps the kernel out of the translation units that include this header for its helpers)
__global__ __launch_bounds__(512, 2) void hv_gemm_wide_kernel(HvGemmParams p, int form) {
    static_assert(BN == 320, "five 64-column blocks per wave");
    constexpr int BM = 256, BK = 64, NS = 2, NW = 8, NB = BN / 64, NMF = 2;
    constexpr int XT = BM * BK * 2, WT = BN * BK * 2, SLOT = XT + WT;
    constexpr int RPI = 8, CPR = 8, RPB = 2;  // rows per 1 KiB wave-instruction, 16-byte chunks per row, rows per bank row
    constexpr int XQ = BM / RPI / NW, WQ = BN / RPI / NW;  // 4 + 5 DMA instructions per wave and k-tile
    __shared__ __attribute__((aligned(16))) unsigned char smem[NS * SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
#ifndef HV_EMU
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#else
    const int wave = tid >> 6;
#endif
    const int r16 = lane & 15, quad = lane >> 4;
    const int tiles_n = p.N / BN, tiles_m = (p.M + BM - 1) / BM;
    const int total = tiles_n * tiles_m;
    const int wg_per_xcd = gridDim.x / 8;
    const int xcd = blockIdx.x % 8, wg = blockIdx.x / 8;
    const int per_xcd = (total + 7) / 8;
    const int t_begin = xcd * per_xcd, t_end = min(total, t_begin + per_xcd);
    const int first = t_begin + wg;
    if (first >= t_end) return;
    const int my_tiles = (t_end - first + wg_per_xcd - 1) / wg_per_xcd;
    const int nk = p.K / BK, nsteps = my_tiles * nk;

    // Issue state.  DMA instruction q of a wave covers tile rows 8 (wave + 8 q) + lane / 8: 64 rows further per q, and both
    // swizzles only look at row bits 0..3 -- so ONE per-lane byte offset per operand serves all q and all k-tiles of a
    // tile (M % 256 == 0, N % 320 == 0: no clamping), and everything that moves (q, k-tile, tile origin) is added to the
    // scalar base.  (Nine per-lane offsets advanced per k-tile, the form of the other kernel, spilled here: the 256 x 320
    // accumulator tile leaves 96 registers.)
    const int sub = lane / CPR, slot = lane & (CPR - 1);
    const int trow0 = RPI * wave + sub;
    const unsigned xofs = ((unsigned)trow0 * (unsigned)p.ldx + (unsigned)((slot ^ ((trow0 / RPB) % CPR)) * 8)) * 2u;
    const unsigned wofs = ((unsigned)trow0 * (unsigned)p.K + (unsigned)((slot ^ hv_wperm_swizzle(trow0)) * 8)) * 2u;
    int i_tile = first, i_k = 0, i_slot = 0;
    auto issue = [&]() __attribute__((always_inline)) {
        unsigned char* sl = smem + i_slot * SLOT;
        const int m0 = (i_tile / tiles_n) * BM, n0 = (i_tile % tiles_n) * BN;
        const char* xb = reinterpret_cast<const char*>(p.X) + ((long)m0 * p.ldx + (long)i_k * BK) * 2;
        const char* wb = reinterpret_cast<const char*>(p.W) + ((long)n0 * p.K + (long)i_k * BK) * 2;
        hv_static_for<XQ>([&](auto Q) __attribute__((always_inline)) {
            constexpr int q = decltype(Q)::value;
            hv_glds16_s(xb + (long)(RPI * NW * q) * p.ldx * 2, xofs, sl + (wave + NW * q) * 1024);
        });
        hv_static_for<WQ>([&](auto Q) __attribute__((always_inline)) {
            constexpr int q = decltype(Q)::value;
            hv_glds16_s(wb + (long)(RPI * NW * q) * p.K * 2, wofs, sl + XT + (wave + NW * q) * 1024);
        });
        if (++i_slot == NS) i_slot = 0;
        if (++i_k == nk) {
            i_k = 0;
            i_tile += wg_per_xcd;
        }
    };

    f32x4 acc[NB][4][NMF];
    auto clear_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < NMF; ++c) acc[b][a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    clear_acc();
    issue();
    int c_tile = first, c_k = 0, c_slot = 0, landed = 0;
    for (int s = 0; s < nsteps; ++s) {
        // (after an epilogue everything issued has landed: it waits for every load before its first store)
        if (landed > 0) --landed;
        else hv_vm_wait<0>();
        hv_barrier_raw();
        const unsigned char* xs = smem + c_slot * SLOT;
        const unsigned char* ws = xs + XT;
        if (++c_slot == NS) c_slot = 0;
        if (s + 1 < nsteps) issue();
        // ten block steps (kk, b): the W fragments of step i+1 are read in front of the eight MFMAs of step i (two register
        // sets, compile-time alternation), and a scheduling fence per step keeps hipcc from hoisting more reads (spills)
        bf16x8 xf[2][NMF], wfa[4], wfb[4];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mf = 0; mf < NMF; ++mf) xf[kk][mf] = hv_as_bf16x8(hv_ld16(xs + hv_swz<BK>(32 * wave + 16 * mf + r16, kk * 4 + quad)));
        auto ldw = [&](int i, bf16x8(&w)[4]) __attribute__((always_inline)) {
            const int kk = i / NB, b = i % NB;
#pragma unroll
            for (int f = 0; f < 4; ++f) w[f] = hv_as_bf16x8(hv_ld16(ws + hv_swz_wperm(64 * b + hv_perm_row(f, r16), kk * 4 + quad)));
        };
        ldw(0, wfa);
        hv_static_for<2 * NB>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value, kk = i / NB, b = i % NB;
            bf16x8(&cur)[4] = (i % 2 == 0) ? wfa : wfb;
            bf16x8(&nxt)[4] = (i % 2 == 0) ? wfb : wfa;
            if constexpr (i + 1 < 2 * NB) ldw(i + 1, nxt);
#pragma unroll
            for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                for (int mf = 0; mf < NMF; ++mf)
                    acc[b][nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[nf], xf[kk][mf], acc[b][nf][mf], 0, 0, 0);
#if !defined(HV_EMU)
            __builtin_amdgcn_sched_barrier(0);
#endif
        });
        if (++c_k == nk) {
            c_k = 0;
            const int m0 = (c_tile / tiles_n) * BM, n0 = (c_tile % tiles_n) * BN;
            {
                const int mw = m0 + 32 * wave;
                const float* tab = p.bias;  // one table row per 32-row wave block (hv_gemm_fast_form(p, 32)); none: the bias, weight 0
                float tscale = 0.f;
                if (p.pe != nullptr) tab = p.pe + (long)((mw / p.pe_period) % p.pe_frames) * p.N, tscale = 1.f;
                else if (p.rowvec != nullptr) tab = p.rowvec + (long)(mw / p.rowvec_period) * p.N, tscale = 1.f;
                if (form == HV_FORM_LN) hv_gemm_epilogue_wide<true, false>(p, acc, mw, n0, r16, quad, tab, tscale);
                else if (form == HV_FORM_RES) hv_gemm_epilogue_wide<false, true>(p, acc, mw, n0, r16, quad, tab, tscale);
                else hv_gemm_epilogue_wide<false, false>(p, acc, mw, n0, r16, quad, tab, tscale);
            }
            landed = 1;
            c_tile += wg_per_xcd;
            clear_acc();
        }
    }
}

#include "hv_gemm4.h"  // the 256 x 256 x 64 tile on four waves of 128 x 128 (round 6)
#include "hv_gemm_xs.h"
#include "hv_gemm_c4.h"
#include "hv_gemm_wr.h"  // X-stationary 192 x 128 tiles for K = 320 (round 6)

static int g_hv_gemm_max_grid = 512;  // tuning knob (hv_set_tuning): persistent workgroups
// tuning knob (hv_set_tuning key 10): which of the problems that take 256 x 256 x 64 tiles run on the four-wave kernel
// (hv_gemm_w4_kernel): 1 (default) = the deferred-store forms at K >= 640 (see hv_gemm_choose), 0 = none (always the 8-wave
// kernel), 2 = every problem whose shape allows it, stores at once, 3 = the same with deferred stores where the form has them
static int g_hv_gemm_w4 = 1;
static int g_hv_gemm_xs = 0;  // tuning knob (hv_set_tuning key 11): 1 = the LayerNorm-fold forms at K = 320 on hv_gemm_xs_kernel
static int g_hv_gemm_w4_units = 1;  // (value 4: as 3 on the plain tile raster instead of the unit raster -- A/B)
// tuning knob (hv_set_tuning key 3) -- kernel selection:
//   1 (default): 256 x 320 x 64 wide tiles for N = 320, K >= 640 (M % 256 == 0, plain-output forms); otherwise 256 x 256 x 64
//      (one 8-wave workgroup per CU) when N >= 960 and its tiles fill the last round over the 256 CUs to >= 90 %, otherwise
//      128 x 128 x 64 (two 4-wave workgroups per CU)
//   2: 256 x 256 x 64 wherever its tile shape is legal (A/Bs), 3: 128 x 128 x 64 everywhere (A/Bs)
//   0: the register-staged kernel for everything (A/Bs; also what problems outside the fast epilogue forms run on)
//   6: as 1 without the wide tiles (the round-3 default; A/B)
static int g_hv_gemm_glds = 1;

// Which kernel hv_gemm_launch takes for a problem: 0 register-staged, 1 = 256x256x64, 2 = 128x128x64 (LDS-DMA), 3 = 256x320x64
// wide tiles, 4 = 256x256x64 on four waves (hv_gemm4.h); perm = the
// permuted channel assignment.  Shared with hv_gemm_gn_parts so that the caller sizes gn_part for the kernel that will run.
struct HvGemmChoice {
    int kernel, form, gm;
    bool perm;
};
// 0: never; 1 (default): deep-K plain / residual projections whose N is a multiple of 320 and that neither the 256-wide
// four-wave tiles nor the statistics-bearing kernels are meant for; 2: wherever the structure allows (tests)
static int g_hv_gemm_c4 = 1;
// hv_gemm_wr_kernel (weights in registers, N = K = 320): 1 = where >= 256 work items of 64 rows exist (default), 0 = never, 2 = always (tests)
static int g_hv_gemm_wr = 1;

static inline HvGemmChoice hv_gemm_choose(const HvGemmParams& p, bool want_stats);

// parts per image of the GroupNorm partial statistics (0 = this problem's kernel cannot emit them)
static inline int hv_gemm_gn_parts_of(const HvGemmParams& p) {
    const HvGemmChoice c = hv_gemm_choose(p, true);
    if (c.kernel == 7) return p.gn_rows_per_image > 0 ? p.gn_rows_per_image / 64 : 0;  // (hv_gemm_choose checked the divisibility)
    if (c.kernel != 2 || !c.perm || (c.form != HV_FORM_RES && c.form != HV_FORM_PLAIN)) return 0;
    if (p.perm_p != 0) return 0;  // (a wave's rows are not one image's after the row permutation)
    const int rows = 64;  // rows of a wave's sub-tile = rows per partial sum
    if (p.gn_rows_per_image <= 0 || p.gn_rows_per_image % rows != 0 || p.M % p.gn_rows_per_image != 0) return 0;
    return p.gn_rows_per_image / rows;
}

static inline HvGemmChoice hv_gemm_choose(const HvGemmParams& p, bool want_stats) {
    HvGemmChoice c{0, HV_FORM_NONE, 1, false};
    const bool prologue = p.pro_scale != nullptr || p.pro_act != HV_ACT_NONE;
    // the LDS-DMA kernel addresses its operands with 32-bit byte offsets and only knows the hot epilogue forms
    const long lim = 1L << 32;
    const bool span_ok = (long)p.M * p.ldx * 2 < lim && (long)p.N * p.K * 2 < lim && (long)p.M * p.ldy * 2 < lim &&
                         (p.X2 == nullptr || (long)p.M * p.ldx2 * 2 < lim) &&
                         (p.residual == nullptr || (long)p.M * p.ldr * 2 < lim) &&
                         (p.Yt == nullptr || (long)(p.N - p.n_split) * p.ldyt * 2 < lim);
    int form128 = hv_gemm_fast_form(p, 128), form64 = hv_gemm_fast_form(p, 64);
    // a positional-encoding table with one row per 16-row fragment (period % 16 == 0 only): the permuted LayerNorm-fold
    // epilogue of the 128 x 128 kernel loads it per fragment (hv_gemm_epilogue_fast_perm)
    if (form64 == HV_FORM_NONE && p.N % 8 == 0 && p.pe != nullptr && hv_gemm_fast_form(p, 16) == HV_FORM_LN)
        form64 = HV_FORM_LN;
    if (!(g_hv_gemm_glds && !prologue && p.M >= 256 && span_ok && form64 != HV_FORM_NONE)) return c;
    // 192 x 320 x 64 tiles on four waves (hv_gemm_c4_kernel): bias (+ residual) outputs without statistics or tables, N a
    // multiple of 320, M of 192.  Default: K >= 1280 with tiles that fill their rounds of 256 CUs to >= 70 % (>= 128 tiles; as
    // hv_conv_w4_width: profiles/r06_s33_gemm_c4_fill.txt), and not where 256-wide tiles fit N (N % 256 == 0:
    // the deferred residual form of hv_gemm_w4_kernel) -- the feed-forward output projections of levels 0 and 1.
    if (g_hv_gemm_c4 && g_hv_gemm_glds != 3 && p.N % 320 == 0 && p.M % 192 == 0 && p.X2 == nullptr && p.perm_p == 0 && p.Yt == nullptr &&
        !p.geglu && !want_stats && p.gn_part == nullptr && p.ln_part == nullptr && p.pe == nullptr && p.rowvec == nullptr) {
        const int form96 = hv_gemm_fast_form(p, 96);
        const long tiles = (long)(p.M / 192) * (p.N / 320);
        if ((form96 == HV_FORM_RES || form96 == HV_FORM_PLAIN) &&
            (g_hv_gemm_c4 == 2 || (p.K >= 1280 && p.N % 256 != 0 && tiles >= 128 && tiles * 10 >= ((tiles + 255) / 256) * 256 * 7))) {
            c.kernel = 6;
            c.form = form96;
            c.perm = true;
            return c;
        }
    }
    // K = 320 with the weights in registers (hv_gemm_wr_kernel): N = 320 bias (+ table row) (+ residual), with or without
    // normalisation statistics of the output; N = 320 / 640 / 960 LayerNorm-fold outputs (+ table row) -- the motion modules' QKV
    if (g_hv_gemm_wr && p.K == 320 && p.N % 320 == 0 && p.N <= 960 && p.M % 64 == 0 && p.X2 == nullptr && p.perm_p == 0 && p.Yt == nullptr &&
        !p.geglu && (g_hv_gemm_wr == 2 || p.M >= 256 * 64) &&
        (p.gn_rows_per_image <= 0 || (p.gn_rows_per_image % 64 == 0 && p.M % p.gn_rows_per_image == 0)) &&
        !(p.gn_part != nullptr && p.ln_part != nullptr)) {
        const bool plain = (form64 == HV_FORM_RES || form64 == HV_FORM_PLAIN) && p.N == 320;
        const bool lnf = form64 == HV_FORM_LN && !want_stats && p.gn_part == nullptr && p.ln_part == nullptr && p.gn_rows_per_image <= 0;
        if (plain || lnf) {
            c.kernel = 7;
            c.form = form64;
            c.perm = true;
            return c;
        }
    }
    // 256 x 320 x 64 wide tiles (hv_gemm_wide_kernel) for N = 320, K >= 640 with a plain-output form on the permuted assignment
    // (level-0 ff2: same-box 0.385 -> 0.327 ms, profiles/r04_s1.txt).  Measured and not taken: K = 320 (0.158 -> 0.159 ms: five
    // k-steps do not amortise the 320-column epilogue), N = 640 (576 tiles at level 1 = 2.25 rounds over the 256 CUs: projection
    // 0.086 -> 0.097 ms, ff2 0.303 -> 0.296).  Tuning value 6 = never (A/B).  The kernel leaves no normalisation statistics: a
    // problem that asks for them stays on the square tiles.
    if (g_hv_gemm_glds != 6 && g_hv_gemm_glds != 2 && g_hv_gemm_glds != 3 && p.N == 320 && p.K >= 640 &&
        p.perm_p == 0 && p.X2 == nullptr && p.M % 256 == 0 && p.Yt == nullptr && !p.geglu && p.bias != nullptr && !want_stats) {
        const int form32 = hv_gemm_fast_form(p, 32);
        if (form32 == HV_FORM_RES || form32 == HV_FORM_PLAIN || form32 == HV_FORM_LN) {
            c.kernel = 3;
            c.form = form32;
            c.perm = true;
            return c;
        }
    }
    const int tm = (p.M + 255) / 256;
    const int n128 = ((p.N + 127) / 128) * 128, n256 = ((p.N + 255) / 256) * 256;
    c.gm = n128 / 128 > 8 ? 8 : 1;  // grouped raster for wide outputs (see the kernel)
    const bool ok128 = form128 != HV_FORM_NONE;  // a per-row table may fit 64-row but not 128-row wave sub-tiles
    // 256 x 256 tiles when N fills them about as well as 128-column tiles would, and when they fill the 256 CUs' last
    // round to >= 90 % (one workgroup per CU: 360 tiles are two rounds at 70 %); otherwise the 128 x 128 x 64 kernel,
    // whose 512 slots quantise four times finer
    const int t256 = tm * (n256 / 256), rounds256 = (t256 + 255) / 256;
    const bool fills256 = t256 * 10 >= rounds256 * 256 * 9;
    const bool shape256 = ok128 && p.N >= 960 && (n256 - p.N) * 8 <= p.N;
    const bool big = g_hv_gemm_glds != 3 && shape256 && (fills256 || g_hv_gemm_glds == 2) &&
                     p.perm_p == 0;  // (value 6 selects like 1 here; a row-permuted output: the 128 x 128 kernel's epilogue)
    c.kernel = big ? 1 : 2;
    c.form = big ? form128 : form64;
    // plain bf16 outputs (plain / residual / LayerNorm fold) and GEGLU with N % 8 == 0: permuted channel assignment, 16-byte epilogue
    c.perm = p.N % 8 == 0 &&
             (c.form == HV_FORM_RES || c.form == HV_FORM_PLAIN || c.form == HV_FORM_LN || c.form == HV_FORM_LN_GEGLU);
    if (p.perm_p != 0 && !c.perm) return HvGemmChoice{0, HV_FORM_NONE, 1, false};  // row-permuted output: that epilogue only
    // the four-wave kernel (hv_gemm4.h):
    //   default (tuning 10 = 1): the deferred-store forms (LayerNorm fold with / without GEGLU, bias + residual; permuted
    //   channels, M % 192 == 0) at K >= 1280 and M >= 16384 where 256-wide tiles fit N -- the class where it wins INSIDE the
    //   step (per-shape step profiles, same box, profiles/r06_s18.txt: level-2 ff1 0.470 -> 0.441 ms per launch, level-2 ff2
    //   (K = 5120, residual in place) 0.258 -> 0.210, level-2 motion QKV 0.198 -> 0.182: -0.93 ms per step).  Not taken:
    //   level 1 (K = 640: ff1 0.474 -> 0.500, QKV 0.206 -> 0.216 in the step although -4 % in isolation), level 3 (M = 4608:
    //   ff1 0.116 -> 0.121), level 0 (K = 320, five k-tiles per tile: the exposed epilogue of a one-wave-per-SIMD kernel
    //   costs more than the deferred stores win: QKV 0.285 -> 0.315, ff1 0.72 -> 0.765);
    //   tuning 10 = 2 / 3 / 4: wherever its shape conditions hold, without / with deferred stores (A/Bs and tests)
    if (g_hv_gemm_w4 && p.N % 64 == 0 && p.X2 == nullptr && p.perm_p == 0 && g_hv_gemm_glds != 3) {
        // (the residual form stores in place: no overlapping last column tile, and no statistics variant of this kernel)
        const bool res_ok = c.form == HV_FORM_RES && p.N % 256 == 0 && !want_stats && p.gn_part == nullptr && p.ln_part == nullptr;
        const bool defer_form = c.perm && (c.form == HV_FORM_LN || c.form == HV_FORM_LN_GEGLU || res_ok) && p.M % 192 == 0 && p.K >= 320 &&
                                hv_gemm_fast_form(p, 96) == c.form;
        // (192-row tiles quantise finer than 256-row ones: level-2 QKV, M = 18 432, N = 3 840, is 5.6 rounds of 192 x 256 tiles
        //  = 94 % where 256 x 256 tiles are 4.2 rounds = 84 % and the selection above takes the 128 x 128 kernel)
        const int t192 = (p.M / 192) * (n256 / 256), rounds192 = (t192 + 255) / 256;
        const bool fills192 = t192 * 10 >= rounds192 * 256 * 9;
        const bool wide = p.N >= 960 && (n256 - p.N) * 8 <= p.N;
        if (g_hv_gemm_w4 == 1) {
            if (defer_form && wide && (big || fills192) && p.K >= 1280 && p.M >= 16384) c.kernel = 4;
        } else if (big && (p.M % 256 == 0 || (g_hv_gemm_w4 == 3 && defer_form))) {
            c.kernel = 4;
        }
    }
    return c;
}

// 64-column blocks per row of the LayerNorm partial statistics (0 = this problem's kernel cannot emit them)
static inline int hv_gemm_ln_parts_of(const HvGemmParams& p) {
    const HvGemmChoice c = hv_gemm_choose(p, true);
    if (c.kernel == 7) return 4;  // one part per wave: 80 columns
    if (c.kernel != 2 || !c.perm || (c.form != HV_FORM_RES && c.form != HV_FORM_PLAIN) || p.N % 64 != 0) return 0;
    return p.N / 64;
}

static inline int hv_gemm_launch(const HvGemmParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return -1;
    if (p.K % 64 != 0 || p.N % 4 != 0) return -1;
    if (p.X2 != nullptr && (p.K1 % 64 != 0)) return -1;
    if (p.geglu && (p.N % 32 != 0 || p.Yt != nullptr || p.out_f32)) return -1;
    if (p.Yt != nullptr && (p.n_split % 16 != 0)) return -1;
    if (p.perm_p != 0 && (p.perm_p < 0 || p.perm_x <= 0 || p.perm_y <= 0 || (long)p.perm_x * p.perm_y * p.perm_p != p.M ||
                          p.Yt != nullptr || p.geglu))
        return -1;
    if (p.gn_part != nullptr && hv_gemm_gn_parts_of(p) == 0) return -1;  // statistics wanted from a kernel that cannot emit them
    if (p.ln_part != nullptr && (hv_gemm_ln_parts_of(p) == 0 || p.gn_part != nullptr)) return -1;
    const bool prologue = p.pro_scale != nullptr || p.pro_act != HV_ACT_NONE;
    char shape[128] = "";
    if (g_hv_prof)
        snprintf(shape, sizeof(shape), "M=%d N=%d K=%d geglu=%d res=%d yt=%d f32=%d x2=%d", p.M, p.N, p.K, p.geglu,
                 p.residual != nullptr, p.Yt != nullptr ? p.N - p.n_split : 0, p.out_f32, p.X2 != nullptr);
    const HvGemmChoice c = hv_gemm_choose(p, p.gn_part != nullptr || p.ln_part != nullptr);
    if (c.kernel == 1) {
        const int t256 = ((p.M + 255) / 256) * ((p.N + 255) / 256);
        int grid = ((t256 + 7) / 8) * 8;
        if (grid > 256) grid = 256;
        if (grid > g_hv_gemm_max_grid) grid = g_hv_gemm_max_grid;
        if (c.perm) {
            hv_note("hv_gemm_glds_kernel<256,8,256,1,perm> | %s", shape);
            hv_launch(hv_gemm_glds_kernel<256, 8, 256, 1, true>, dim3(grid), dim3(512), stream, p, c.gm, c.form);
        } else {
            hv_note("hv_gemm_glds_kernel<256,8,256,1> | %s", shape);
            hv_launch(hv_gemm_glds_kernel<256, 8, 256, 1, false>, dim3(grid), dim3(512), stream, p, c.gm, c.form);
        }
        return 0;
    }
    if (c.kernel == 7) {
        const int items = p.M / 64;
        const int grid = 8 * min(min(32, g_hv_gemm_max_grid / 8), max((items + 7) / 8, p.N / 320));
        if (c.form == HV_FORM_LN) {
            hv_note("hv_gemm_wr_kernel<lnfold> | %s", shape);
            hv_launch(hv_gemm_wr_kernel<0, true>, dim3(grid), dim3(256), stream, p);
        } else if (p.gn_part != nullptr) {
            hv_note("hv_gemm_wr_kernel<gn> | %s", shape);
            hv_launch(hv_gemm_wr_kernel<1>, dim3(grid), dim3(256), stream, p);
        } else if (p.ln_part != nullptr) {
            hv_note("hv_gemm_wr_kernel<ln> | %s", shape);
            hv_launch(hv_gemm_wr_kernel<2>, dim3(grid), dim3(256), stream, p);
        } else {
            hv_note("hv_gemm_wr_kernel | %s", shape);
            hv_launch(hv_gemm_wr_kernel<0>, dim3(grid), dim3(256), stream, p);
        }
        return 0;
    }
    if (c.kernel == 6) {
        const int tiles = (p.M / 192) * (p.N / 320);
        hv_note("hv_gemm_c4_kernel | %s", shape);
        hv_launch(hv_gemm_c4_kernel<0>, dim3(((tiles + 7) / 8) * 8), dim3(256), stream, p);
        return 0;
    }
    if (c.kernel == 5) {
        const int tiles_m = p.M / 192, tiles_n = (p.N + 127) / 128;
        auto grid_of = [&](int work) {
            int g = ((work + 7) / 8) * 8;
            if (g > 256) g = 256;
            if (g > g_hv_gemm_max_grid) g = g_hv_gemm_max_grid;
            return g;
        };
        auto fill = [&](int work) {
            const int g = grid_of(work), per_xcd = (work + 7) / 8, rounds = (per_xcd + g / 8 - 1) / (g / 8);
            return (double)work / ((double)rounds * g);
        };
        // column tiles per unit: the largest divisor of tiles_n whose units fill the workgroups' rounds (the resident X block
        // is loaded once per unit)
        int U = 1;
        const double f1 = fill(tiles_m * tiles_n);
        for (int u = tiles_n; u > 1; --u)
            if (tiles_n % u == 0 && fill(tiles_m * (tiles_n / u)) >= 0.97 * f1) {
                U = u;
                break;
            }
        const int grid = grid_of(tiles_m * (tiles_n / U));
        hv_note("hv_gemm_xs_kernel | %s", shape);
        if (c.form == HV_FORM_LN) hv_launch(hv_gemm_xs_kernel<1>, dim3(grid), dim3(256), stream, p, U);
        else hv_launch(hv_gemm_xs_kernel<2>, dim3(grid), dim3(256), stream, p, U);
        return 0;
    }
    if (c.kernel == 4) {
        // 192-row tiles (waves of 96 x 128) with deferred output stores for the two forms that carry the volume (K >= 320:
        // four k-tiles of the next tile take them); 256-row tiles with the epilogues of this file otherwise
        const bool defer = g_hv_gemm_w4 != 2 && c.perm && p.K >= 320 && p.M % 192 == 0 &&
                           (c.form == HV_FORM_LN || c.form == HV_FORM_LN_GEGLU ||
                            (c.form == HV_FORM_RES && p.N % 256 == 0 && p.gn_part == nullptr && p.ln_part == nullptr)) &&
                           hv_gemm_fast_form(p, 96) == c.form;
        const int bm = defer ? 192 : 256;
        const int tiles_m = p.M / bm, tiles_n = (p.N + 255) / 256;
        const int tiles = tiles_m * tiles_n;
        auto grid_of = [&](int work) {
            int g = ((work + 7) / 8) * 8;
            if (g > 256) g = 256;
            if (g > g_hv_gemm_max_grid) g = g_hv_gemm_max_grid;
            return g;
        };
        // Unit raster (see the kernel) for the streamed-X shapes (K <= 640: a row block of X is 123 / 246 KB): the largest
        // number U of column tiles per unit (a divisor of tiles_n) that fills the workgroups' rounds as well as single tiles do
        int gm = c.gm, grid = grid_of(tiles);
        if (defer && p.K <= 640 && g_hv_gemm_w4_units) {
            auto fill = [&](int work) {
                const int g = grid_of(work), per_xcd = (work + 7) / 8, rounds = (per_xcd + g / 8 - 1) / (g / 8);
                return (double)work / ((double)rounds * g);
            };
            const double f1 = fill(tiles);
            for (int U = tiles_n; U > 1; --U) {
                if (tiles_n % U != 0) continue;
                const int units = tiles_m * (tiles_n / U);
                if (fill(units) >= 0.97 * f1) {
                    gm = -U;
                    grid = grid_of(units);
                    break;
                }
            }
        }
        if (defer && c.form == HV_FORM_LN) {
            hv_note("hv_gemm_w4_kernel<perm,defer,192> | %s", shape);
            hv_launch(hv_gemm_w4_kernel<true, 1, 6>, dim3(grid), dim3(256), stream, p, gm, c.form);
        } else if (defer && c.form == HV_FORM_RES) {
            hv_note("hv_gemm_w4_kernel<perm,defer,192> | %s", shape);
            hv_launch(hv_gemm_w4_kernel<true, 3, 6>, dim3(grid), dim3(256), stream, p, gm, c.form);
        } else if (defer) {
            hv_note("hv_gemm_w4_kernel<perm,defer,192> | %s", shape);
            hv_launch(hv_gemm_w4_kernel<true, 2, 6>, dim3(grid), dim3(256), stream, p, gm, c.form);
        } else if (c.perm) {
            hv_note("hv_gemm_w4_kernel<perm> | %s", shape);
            hv_launch(hv_gemm_w4_kernel<true, 0, 8>, dim3(grid), dim3(256), stream, p, c.gm, c.form);
        } else {
            hv_note("hv_gemm_w4_kernel | %s", shape);
            hv_launch(hv_gemm_w4_kernel<false, 0, 8>, dim3(grid), dim3(256), stream, p, c.gm, c.form);
        }
        return 0;
    }
    if (c.kernel == 3) {
        const int tw = ((p.M + 255) / 256) * (p.N / 320);
        int grid = ((tw + 7) / 8) * 8;
        if (grid > 256) grid = 256;
        if (grid > g_hv_gemm_max_grid) grid = g_hv_gemm_max_grid;
        hv_note("hv_gemm_wide_kernel | %s", shape);
        hv_launch(hv_gemm_wide_kernel<320>, dim3(grid), dim3(512), stream, p, c.form);
        return 0;
    }
    if (c.kernel == 2) {
        const int tiles6 = ((p.M + 127) / 128) * ((p.N + 127) / 128);
        int grid6 = ((tiles6 + 7) / 8) * 8;
        if (grid6 > 512) grid6 = 512;
        if (grid6 > g_hv_gemm_max_grid) grid6 = g_hv_gemm_max_grid;
        if (c.perm && p.gn_part != nullptr) {
            hv_note("hv_gemm_glds_kernel<128,4,128,2,perm,gn> | %s", shape);
            hv_launch(hv_gemm_glds_kernel<128, 4, 128, 2, true, 1>, dim3(grid6), dim3(256), stream, p, c.gm, c.form);
        } else if (c.perm && p.ln_part != nullptr) {
            hv_note("hv_gemm_glds_kernel<128,4,128,2,perm,ln> | %s", shape);
            hv_launch(hv_gemm_glds_kernel<128, 4, 128, 2, true, 2>, dim3(grid6), dim3(256), stream, p, c.gm, c.form);
        } else if (c.perm) {
            hv_note("hv_gemm_glds_kernel<128,4,128,2,perm> | %s", shape);
            hv_launch(hv_gemm_glds_kernel<128, 4, 128, 2, true>, dim3(grid6), dim3(256), stream, p, c.gm, c.form);
        } else {
            hv_note("hv_gemm_glds_kernel<128,4,128,2> | %s", shape);
            hv_launch(hv_gemm_glds_kernel<128, 4, 128, 2, false>, dim3(grid6), dim3(256), stream, p, c.gm, c.form);
        }
        return 0;
    }
    const int tiles = ((p.N + 127) / 128) * ((p.M + 127) / 128);
    // persistent grid: 2 workgroups per CU (64 KiB LDS each), 256 CUs, fewer when the problem is small
    int grid = ((tiles + 7) / 8) * 8;
    if (grid > g_hv_gemm_max_grid) grid = g_hv_gemm_max_grid;
    hv_note("hv_gemm_kernel<64>%s | %s", prologue ? "+gn_prologue" : "", shape);
    hv_launch(hv_gemm_kernel<64>, dim3(grid), dim3(256), stream, p);
    return 0;
}
