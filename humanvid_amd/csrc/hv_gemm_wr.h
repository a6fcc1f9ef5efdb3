// hv_gemm_wr.h -- Y = X . W^T + bias (+ residual) for N = K = 320 with the WEIGHTS IN REGISTERS: the level-0 output projections
// (attention to_out / proj_out of the spatial transformers and motion modules, /root/reference/src/models/transformer_3d.py:125-166,
// motion_module.py:157-175) are HBM streams with 60 GFLOP attached -- 566 MB per launch at M = 294 912 against 0.03 ms of MFMA
// work -- and the tile kernels spend their time in per-tile set-up and epilogue rounds (3.5 TB/s).  Here nothing is per tile:
//   * a persistent workgroup of four waves loads W ONCE: wave w owns output channels 80 w .. 80 w + 79 as 5 x 10 A-fragments
//     (200 registers: 120 in the accumulation file, 80 vector), for the whole launch;
//   * X and the residual stream through two 4-slot LDS rings of 32-row units (20 KiB each, rows of 128 bytes per 64-channel
//     chunk, the GEMM's swizzle) by LDS-DMA, three units ahead: 120 KiB in flight per CU; every wave reads every X fragment
//     (20 ds_read_b128 per unit and wave beside 100 MFMAs) and its own channels of the residual;
//   * per unit: one barrier, 100 MFMAs per wave, the epilogue (bias, residual from LDS, statistics, 16-byte stores) -- the next
//     units' copies are in flight throughout.
// Statistics of the output for the normalisation that follows (STATS): 1 = GroupNorm parts per 64 rows (two units) and channel,
// the layout of hv_gemm_epilogue_fast_perm<., ., ., 1>; 2 = LayerNorm parts per row and WAVE (four parts of 80 columns:
// hv_gemm_ln_parts() reports 4 for this kernel).  Accumulation runs over ascending k from zero and the output arithmetic is
// (acc + bias) + residual: Y is bit-identical to the other kernel selections; the statistics differ in summation order only.
// N > 320 (640, 960): a workgroup keeps ONE 320-column tile of W; the workgroups of an XCD that hold different tiles walk the
// same rows, so X comes from HBM once and from L2 for the others.  LN = true: the LayerNorm-fold form
// y = rstd[m] (acc - mean[m] colsum[n]) + bias[n] (+ table row) of the QKV projections (the rows' mean / rstd ride the residual
// ring's LDS by two small copies per unit).  A per-row-block table (positional-encoding row per frame / row vector per batch:
// hv_gemm_params pe / rowvec) is added to the bias; it is re-read when the unit's block changes -- an ordinary load, i.e. one
// drain of the copies in flight per change (every 192 units at 6144 rows per frame).
#pragma once
#include "hv_common.h"
#include "hv_gemm4.h"  // hv_glds16_u, hv_acc_take, hv_acc_settle, hv_mfma_tied
#include "humanvid_hip.h"

struct HvGemmWrGeom {
    static constexpr int N = 320, K = 320, UR = 32;      // unit: 32 rows
    static constexpr int UNIT_B = 5 * UR * 128;           // 20 480 bytes: [chunk 5][row 32][128 B]
    static constexpr int RING = 4, R0 = RING * UNIT_B, LDS_B = 2 * RING * UNIT_B;  // X ring, residual ring: 160 KiB
};

// acc += A . B with A in the accumulation file (gfx90a+: MFMA source operands may come from either file)
HV_DEV void hv_mfma_tied_a(f32x4& acc, const bf16x8& a, const bf16x8& b) {
#ifndef HV_EMU
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "v"(b));
#else
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
#endif
}

template <int STATS, bool LN = false>
__global__ __launch_bounds__(256, 1) void hv_gemm_wr_kernel(hv_gemm_params p) {
    using G = HvGemmWrGeom;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[G::LDS_B];
    const int tid = threadIdx.x, lane = tid & 63;
#ifndef HV_EMU
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#else
    const int wave = tid >> 6;
#endif
    const int r16 = lane & 15, quad = lane >> 4;
    const bool has_res = !LN && p.residual != nullptr;

    // work items of 64 rows (two units): XCD x owns a contiguous range of items; its gridDim / 8 workgroups are dealt round-robin
    // to the N / 320 column tiles, and the workgroups of one tile take every (their count)-th item of the XCD's range
    const int tiles_n = p.N / G::N;
    const int items = p.M / 64, nwg = gridDim.x / 8, per_xcd = (items + 7) / 8;
    // (the same number of workgroups per tile -- 32 per XCD: 10 + 10 + 10 at three tiles, two idle --, so that the workgroups that hold
    //  different tiles reach the same rows at the same time; dealing all 32 as 11 / 11 / 10 measured the same, profiles/r06_s36 / s37:
    //  at three tiles the kernel is not bound by X -- per unit 100 MFMAs and the LayerNorm-fold epilogue run one after the other on
    //  each SIMD's single wave, 2.2 us per unit against 0.8 us of MFMAs)
    const int q = (int)(blockIdx.x / 8);
    const int stride = nwg / tiles_n;  // workgroups of this XCD per tile
    if (q >= stride * tiles_n) return;
    const int nt = q % tiles_n, n0 = nt * G::N;
    const int i_end = min(items, ((int)(blockIdx.x % 8) + 1) * per_xcd);
    const int i_first = (int)(blockIdx.x % 8) * per_xcd + q / tiles_n;
    if (i_first >= i_end) return;
    const int n_items = (i_end - i_first + stride - 1) / stride;
    const int n_units = 2 * n_items;
    auto unit_row = [&](int u) __attribute__((always_inline)) {  // first row of unit u of this workgroup (clamped past the end)
        const int uu = min(u, n_units - 1);
        return (i_first + (uu >> 1) * stride) * 64 + (uu & 1) * G::UR;
    };

    // ---- W into registers: fragment nf of the wave, k step kc: lane (row r16 of the fragment, k = 32 kc + 8 quad ..)
    // channel of (nf, r): pairs (0, 1), (2, 3) as hv_perm_row (a lane's D rows of a pair are eight consecutive channels), nf = 4 plain
    bf16x8 wa[3][10], wv[2][10];
    {
        const uint16_t* wsrc = reinterpret_cast<const uint16_t*>(p.W);
#pragma unroll
        for (int nf = 0; nf < 5; ++nf) {
            const int ch = n0 + 80 * wave + (nf < 4 ? 32 * (nf >> 1) + 8 * (r16 >> 2) + 4 * (nf & 1) + (r16 & 3) : 64 + r16);
#pragma unroll
            for (int kc = 0; kc < 10; ++kc) {
                const bf16x8 f = hv_as_bf16x8(hv_ld16(wsrc + (long)ch * G::K + 32 * kc + 8 * quad));
                if (nf < 3) wa[nf][kc] = f;
                else wv[nf - 3][kc] = f;
            }
        }
    }

    // ---- LDS-DMA: a unit is 20 instructions per operand; wave w carries rows 8 w .. 8 w + 7 of every chunk j = 0..4
    const int key = (4 * wave + (lane >> 4)) & 7;  // (row >> 1) & 7 of the lane's row 8 w + (l >> 3)
    const unsigned xofs = ((unsigned)(8 * wave + (lane >> 3)) * (unsigned)p.ldx + (unsigned)(((lane & 7) ^ key) * 8)) * 2u;
    const unsigned rofs = ((unsigned)(8 * wave + (lane >> 3)) * (unsigned)p.ldr + (unsigned)(((lane & 7) ^ key) * 8)) * 2u;
    auto issue_unit = [&](int u) __attribute__((always_inline)) {
        const long m0 = unit_row(u);
        const unsigned slot = (unsigned)(u & 3) * (unsigned)G::UNIT_B;
        const char* xs = reinterpret_cast<const char*>(p.X) + m0 * p.ldx * 2;
#pragma unroll
        for (int j = 0; j < 5; ++j) hv_glds16_u(xs + j * 128, xofs, smem + slot + j * 4096 + wave * 1024);
        if (has_res) {  // (wave-uniform)
            const char* rs = reinterpret_cast<const char*>(p.residual) + m0 * p.ldr * 2;
#pragma unroll
            for (int j = 0; j < 5; ++j) hv_glds16_u(rs + j * 128, rofs, smem + G::R0 + slot + j * 4096 + wave * 1024);
        }
        if (LN) {  // the unit's 32 row means and 32 row rstds: 128 bytes each, lanes 0-7 (every wave copies them: the same bytes)
#ifndef HV_EMU
            const unsigned long m8 = 0xfful;
#else
            const unsigned long m8 = lane < 8 ? 1ul : 0ul;  // (emulator: the lane's own bit)
#endif
            hv_glds16_um(p.row_mean + m0, (unsigned)lane * 16u, smem + G::R0 + slot, m8);
            hv_glds16_um(p.row_rstd + m0, (unsigned)lane * 16u, smem + G::R0 + slot + 128, m8);
        }
    };
    const int per_unit = LN ? 7 : (has_res ? 10 : 5);  // copies per wave and unit

    // fragment / residual read offsets inside a unit: row 16 mf + r16, chunk c, piece q: c * 4096 + row * 128 + ((q ^ key_r) << 4)
    const unsigned key_r = (unsigned)((r16 >> 1) & 7);
    const unsigned xrow = (unsigned)r16 * 128u;
    auto x_frag = [&](unsigned slot, int mf, int kc) __attribute__((always_inline)) {
        const unsigned a = slot + (unsigned)(kc >> 1) * 4096u + (unsigned)mf * 2048u + xrow + (((unsigned)(4 * (kc & 1) + quad) ^ key_r) << 4);
        return hv_as_bf16x8(hv_ld16(smem + a));
    };
    // the lane's residual pieces: pair pp (0, 1): channels 80 w + 32 pp + 8 quad .. + 7; single: 80 w + 64 + 4 quad .. + 3
    int c_pair[2], c_one;
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) c_pair[pp] = 80 * wave + 32 * pp + 8 * quad;  // (inside the 320-column tile: + n0 in global memory)
    c_one = 80 * wave + 64 + 4 * quad;
    auto r_addr = [&](unsigned slot, int mf, int c) __attribute__((always_inline)) {  // byte address of channel c (multiple of 4)
        return G::R0 + slot + (unsigned)(c >> 6) * 4096u + (unsigned)mf * 2048u + xrow + ((((unsigned)(c & 63) >> 3) ^ key_r) << 4) +
               (unsigned)(c & 7) * 2u;
    };

    // bias (+ the table row of the current row block) and, LN, the column sums of the lane's channels
    f32x4 b_pair[2][2], b_one, a_pair[2][2], a_one, cs_pair[2][2], cs_one;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            b_pair[pp][h] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n0 + c_pair[pp] + 4 * h) : zero4;
            cs_pair[pp][h] = LN ? *reinterpret_cast<const f32x4*>(p.colsum + n0 + c_pair[pp] + 4 * h) : zero4;
            a_pair[pp][h] = b_pair[pp][h];
        }
    b_one = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n0 + c_one) : zero4;
    cs_one = LN ? *reinterpret_cast<const f32x4*>(p.colsum + n0 + c_one) : zero4;
    a_one = b_one;
    const float* const table = p.pe != nullptr ? p.pe : p.rowvec;  // (hv_gemm_fast_form: never both)
    const int t_period = p.pe != nullptr ? p.pe_period : p.rowvec_period;
    int t_row = -1;

    // ---- prologue: units 0, 1, 2 in flight; W and bias loads done (compiler-visible: waited where they are used)
    issue_unit(0);
    issue_unit(1);
    issue_unit(2);

    f32x4 acc[2][5];
    f32x4 gs_pair[2][2], gq_pair[2][2], gs_one, gq_one;  // STATS == 1: per-channel sums over the 64 rows of a work item
    uint16_t* const yb = reinterpret_cast<uint16_t*>(p.Y);
    for (int u = 0; u < n_units; ++u) {
        // unit u landed (the two units behind it may stay in flight), every wave done with the slot of unit u - 1
        if (per_unit == 10) hv_vm_wait<20>();
        else if (per_unit == 7) hv_vm_wait<14>();
        else hv_vm_wait<10>();
        hv_barrier_raw();
        issue_unit(u + 3);  // (past the end: the last unit again, into a slot nobody reads)
        const unsigned slot = (unsigned)(u & 3) * (unsigned)G::UNIT_B;
        const long m0 = unit_row(u);
        if (table != nullptr) {  // (wave-uniform) the unit's table row; re-read only when it changes
            int tr = (int)(m0 / t_period);
            if (p.pe != nullptr) tr %= p.pe_frames;
            if (tr != t_row) {
                t_row = tr;
                const float* trow = table + (long)tr * p.N + n0;
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        a_pair[pp][h] = b_pair[pp][h] + *reinterpret_cast<const f32x4*>(trow + c_pair[pp] + 4 * h);
                a_one = b_one + *reinterpret_cast<const f32x4*>(trow + c_one);
            }
        }
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int nf = 0; nf < 5; ++nf) acc[mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 xb[2][2];
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) xb[0][mf] = x_frag(slot, mf, 0);
#pragma unroll
        for (int kc = 0; kc < 10; ++kc) {
            if (kc + 1 < 10) {
#pragma unroll
                for (int mf = 0; mf < 2; ++mf) xb[(kc + 1) & 1][mf] = x_frag(slot, mf, kc + 1);
            }
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
#pragma unroll
                for (int nf = 0; nf < 3; ++nf) hv_mfma_tied_a(acc[mf][nf], wa[nf][kc], xb[kc & 1][mf]);
#pragma unroll
                for (int nf = 3; nf < 5; ++nf) hv_mfma_tied(acc[mf][nf], wv[nf - 3][kc], xb[kc & 1][mf]);
            }
        }
        // ---- epilogue of the unit
        hv_acc_settle();
        if (STATS == 1 && (u & 1) == 0) {
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int h = 0; h < 2; ++h) gs_pair[pp][h] = gq_pair[pp][h] = f32x4{0.f, 0.f, 0.f, 0.f};
            gs_one = gq_one = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
            const long m = m0 + 16 * mf + r16;
            float rs = 0.f, rq = 0.f;  // STATS == 2: the lane's share of its row
            float mean = 0.f, rstd = 1.f;
            if (LN) {
                mean = *reinterpret_cast<const float*>(smem + G::R0 + slot + (unsigned)(16 * mf + r16) * 4u);
                rstd = *reinterpret_cast<const float*>(smem + G::R0 + slot + 128u + (unsigned)(16 * mf + r16) * 4u);
            }
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                f32x4 v0 = hv_acc_take(acc[mf][2 * pp]), v1 = hv_acc_take(acc[mf][2 * pp + 1]);
                if (LN) v0 = rstd * (v0 - mean * cs_pair[pp][0]), v1 = rstd * (v1 - mean * cs_pair[pp][1]);
                v0 += a_pair[pp][0];
                v1 += a_pair[pp][1];
                if (has_res) {
                    const u32x4 r = hv_ld16(smem + r_addr(slot, mf, c_pair[pp]));
                    v0 += f32x4{hv_bf2f((bf16_t)(r[0] & 0xffff)), hv_bf2f((bf16_t)(r[0] >> 16)), hv_bf2f((bf16_t)(r[1] & 0xffff)),
                                hv_bf2f((bf16_t)(r[1] >> 16))};
                    v1 += f32x4{hv_bf2f((bf16_t)(r[2] & 0xffff)), hv_bf2f((bf16_t)(r[2] >> 16)), hv_bf2f((bf16_t)(r[3] & 0xffff)),
                                hv_bf2f((bf16_t)(r[3] >> 16))};
                }
                hv_st16(yb + m * p.ldy + n0 + c_pair[pp],
                        u32x4{hv_pack2(v0[0], v0[1]), hv_pack2(v0[2], v0[3]), hv_pack2(v1[0], v1[1]), hv_pack2(v1[2], v1[3])});
                if (STATS == 1) {
                    gs_pair[pp][0] += v0, gq_pair[pp][0] += v0 * v0;
                    gs_pair[pp][1] += v1, gq_pair[pp][1] += v1 * v1;
                }
                if (STATS == 2) {
                    rs += ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]));
                    rq += ((v0[0] * v0[0] + v0[1] * v0[1]) + (v0[2] * v0[2] + v0[3] * v0[3])) +
                          ((v1[0] * v1[0] + v1[1] * v1[1]) + (v1[2] * v1[2] + v1[3] * v1[3]));
                }
            }
            {
                f32x4 v = hv_acc_take(acc[mf][4]);
                if (LN) v = rstd * (v - mean * cs_one);
                v += a_one;
                if (has_res) {
                    const u32x2 r = hv_ld8(smem + r_addr(slot, mf, c_one));
                    v += f32x4{hv_bf2f((bf16_t)(r[0] & 0xffff)), hv_bf2f((bf16_t)(r[0] >> 16)), hv_bf2f((bf16_t)(r[1] & 0xffff)),
                               hv_bf2f((bf16_t)(r[1] >> 16))};
                }
                hv_st8(yb + m * p.ldy + n0 + c_one, u32x2{hv_pack2(v[0], v[1]), hv_pack2(v[2], v[3])});
                if (STATS == 1) gs_one += v, gq_one += v * v;
                if (STATS == 2) {
                    rs += (v[0] + v[1]) + (v[2] + v[3]);
                    rq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                }
            }
            if (STATS == 2) {  // the row's sum over the wave's 80 columns: over the four quads
                rs += __shfl_xor(rs, 16);
                rq += __shfl_xor(rq, 16);
                rs += __shfl_xor(rs, 32);
                rq += __shfl_xor(rq, 32);
                if (quad == 0)
                    *reinterpret_cast<u32x2*>(p.ln_part + (m * 4 + wave) * 2) =
                        u32x2{__builtin_bit_cast(unsigned, rs), __builtin_bit_cast(unsigned, rq)};
            }
        }
        if (STATS == 1 && (u & 1) == 1) {  // the work item's 64 rows are one GroupNorm part: over the 16 lanes of the DPP row
            const long mi = m0 - G::UR;    // first row of the item
            const int parts = p.gn_rows_per_image / 64;
            const int img = (int)(mi / p.gn_rows_per_image), part = (int)((mi - (long)img * p.gn_rows_per_image) / 64);
            float* dst = p.gn_part + ((long)img * parts + part) * G::N * 2;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 a, b;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a[e] = hv_row16_sum(gs_pair[pp][h][e]);
                        b[e] = hv_row16_sum(gq_pair[pp][h][e]);
                    }
                    if (r16 == 0) {
                        const int n = c_pair[pp] + 4 * h;
                        *reinterpret_cast<f32x4*>(dst + 2 * n) = f32x4{a[0], b[0], a[1], b[1]};
                        *reinterpret_cast<f32x4*>(dst + 2 * n + 4) = f32x4{a[2], b[2], a[3], b[3]};
                    }
                }
            f32x4 a, b;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] = hv_row16_sum(gs_one[e]);
                b[e] = hv_row16_sum(gq_one[e]);
            }
            if (r16 == 0) {
                *reinterpret_cast<f32x4*>(dst + 2 * c_one) = f32x4{a[0], b[0], a[1], b[1]};
                *reinterpret_cast<f32x4*>(dst + 2 * c_one + 4) = f32x4{a[2], b[2], a[3], b[3]};
            }
        }
    }
    hv_vm_wait<0>();  // the copies past the end: landed before this workgroup's LDS is released
}
