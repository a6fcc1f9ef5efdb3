// hv_gemm4.h -- the 256 x 256 x 64 GEMM tile on FOUR waves of 128 x 128 (round 6).  Included by hv_gemm.h (it uses the
// epilogues defined there); serves the wide LayerNorm-fold outputs of the denoising path: fused QKV projections
// (src/models/mutual_self_attention.py:147-186, src/models/motion_module.py:233-256) and the GEGLU input projection of every
// feed-forward (diffusers FeedForward, src/models/attention.py:427).
//
// Why a second 256 x 256 kernel.  The 8-wave kernel (hv_gemm_glds_kernel<256, 8, 256, 1>) puts two waves of 128 x 64 on
// each SIMD: 256 registers per wave, all of them taken by 128 accumulators + fragments, its eight waves re-synchronised
// twice per k-tile, and a tile's 64 - 128 KiB of output stores issued in one burst that the next counted vmcnt wait has to
// see acknowledged (stores retire in order with the LDS-DMA on gfx9) -- profiles/r05_s12_gemm_epilogue_split.txt: a launch
// costs k-loop + epilogue at every shape.  Here ONE wave per SIMD owns a 128 x 128 sub-tile:
//   * 512 registers per wave: 256 accumulators, the two fragment sets, and room to keep a finished tile's packed
//     bf16 results (128 registers) while the NEXT tile is being multiplied -- the results are stored a few at a time behind
//     the next tile's LDS-DMA pieces, so no wait ever covers a fresh burst of stores;
//   * half the LDS fragment bytes per MFMA (one 16-byte fragment read per 4 MFMAs instead of per 2.67);
//   * one raw barrier per k-tile between four waves; the barrier sits AFTER the last fragment read of a k-tile, so the
//     whole slot is free behind it: X runs three k-tiles ahead on a 3-slot ring and W two on a 2-slot ring (160 KiB);
//   * the k-tile is written as 16 blocks of [2 fragment reads, 1 LDS-DMA piece, (1 store), 8 MFMAs] in source order with a
//     scheduling fence per block: every memory instruction issues in the shadow of the 8 MFMAs (128 matrix cycles) of
//     its block, fragments are read two blocks ahead of their MFMAs.
// MFMA order per accumulator is k-ascending as in every other GEMM kernel of this file: results are bit-identical to them.
#pragma once

#ifndef HV_EMU
// one LDS-DMA piece (64 lanes x 16 B): source = wave-uniform base (SGPR pair) + per-lane 32-bit byte offset, destination =
// wave-uniform LDS byte address through M0 (written in the same statement; this kernel has no other M0 user).  Invisible to
// hipcc's wait counts: completion is tracked by hand (hv_vm_wait).
HV_DEV void hv_glds16_u(const void* base_uniform, unsigned byte_ofs, void* lds_wave_base) {
    const unsigned lds_addr_uniform = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)lds_wave_base;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(byte_ofs), "s"(base_uniform), "s"(lds_addr_uniform)
                 : "memory");
}
#else
HV_DEV void hv_glds16_u(const void* base_uniform, unsigned byte_ofs, void* lds_wave_base) {
    memcpy((char*)lds_wave_base + (threadIdx.x & 63) * 16, (const char*)base_uniform + byte_ofs, 16);
}
#endif


// one LDS-DMA piece with the lanes outside `mask` switched off (they neither load nor write LDS).  The kernel runs with all
// 64 lanes active wherever this is called: EXEC is restored to -1.
#ifndef HV_EMU
HV_DEV void hv_glds16_um(const void* base_uniform, unsigned byte_ofs, void* lds_wave_base, unsigned long mask) {
    const unsigned lds_addr_uniform = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)lds_wave_base;
    asm volatile("s_mov_b64 exec, %3\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1"
                 :
                 : "v"(byte_ofs), "s"(base_uniform), "s"(lds_addr_uniform), "s"(mask)
                 : "memory");
}
HV_DEV unsigned long hv_lane_mask(bool on) { return __builtin_amdgcn_ballot_w64(on); }
#else
HV_DEV void hv_glds16_um(const void* base_uniform, unsigned byte_ofs, void* lds_wave_base, unsigned long mask) {
    if (mask) memcpy((char*)lds_wave_base + (threadIdx.x & 63) * 16, (const char*)base_uniform + byte_ofs, 16);
}
HV_DEV unsigned long hv_lane_mask(bool on) { return on ? 1ul : 0ul; }  // emulator: the lane's own bit
#endif


// One accumulator fragment for the vector ALU, read where it is consumed.  The accumulators live in the accumulation registers
// (a0..); VALU instructions cannot read those, and left to itself hipcc's allocator copies ALL of a tile's accumulators into
// v-registers at the head of the k-tile that carries the epilogue (128 registers held through the epilogue's loads: the
// packed results of the previous groups were then spilled to scratch, and every scratch reload waits vmcnt(0) in the middle
// of a k-tile, draining the LDS-DMA ring -- measured 2 x the time of the whole launch).  The "a" constraint keeps the
// fragment in the accumulation file up to this statement.  hipcc pads no hazards around an asm statement: the caller leaves
// the MFMA -> v_accvgpr_read wait states before the first call (hv_acc_settle).
HV_DEV f32x4 hv_acc_take(const f32x4& a) {
#ifndef HV_EMU
    float x0, x1, x2, x3;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x0) : "a"(a[0]));
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x1) : "a"(a[1]));
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x2) : "a"(a[2]));
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x3) : "a"(a[3]));
    return f32x4{x0, x1, x2, x3};
#else
    return a;
#endif
}
HV_DEV void hv_acc_settle() {
#ifndef HV_EMU
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // 32 wait states: more than any MFMA -> accumulator-read hazard
#endif
}

// acc += A . B with the accumulator TIED to its accumulation registers.  With the builtin hipcc's allocator takes the untied
// form for a third of hv_conv_w4_kernel's 1 080 MFMAs per loop body (240 accumulators in 256 registers leave it room to) and then
// permutes the accumulators back at the loop head: ~630 v_accvgpr moves per nine k-tiles.  An asm statement gets no hazard
// padding from hipcc: operands from ds_read are covered by the s_waitcnt the compiler still places in front of the statement,
// an accumulator is touched once per 60 MFMAs, and the epilogue leaves the read-after-MFMA wait states itself (hv_acc_settle).
HV_DEV void hv_mfma_tied(f32x4& acc, const bf16x8& a, const bf16x8& b) {
#ifndef HV_EMU
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
#else
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
#endif
}

// ---- epilogue of the deferred forms, in two phases so that ALL of a tile's operand loads are one round trip: the trace
// build (profiles/r06_s4_w4_trace.txt) showed the epilogue at 5 800 - 8 200 cycles per tile with the loads requested group by
// group (six dependent L2 round trips behind the in-flight LDS-DMA).
// Phase 1 (hv_gemm4_load_cols / the row statistics in pack_tile): the per-column vectors of one 64-channel block under the
// form's channel assignment -- add = bias (+ table row), cs = column sums of the folded weight -- and the rows' mean / rstd.
// Phase 2 (hv_gemm4_pack_*): the arithmetic, results left PACKED in registers instead of stored.  Operation for operation
// the arithmetic of hv_gemm_epilogue_fast_perm<NMF, true, false> / hv_gemm_epilogue_fast_perm_geglu<NMF> (hv_gemm.h):
// bit-identical outputs.
struct HvGemm4Cols {
    f32x4 add[4], cs[4], tab[4];
};
template <bool GEGLU, bool LN = true>
HV_DEV void hv_gemm4_load_cols(const HvGemmParams& p, int n_base, int quad, const float* tab_row, HvGemm4Cols& c) {
    auto ld4 = [&](const float* base, unsigned byte_ofs) __attribute__((always_inline)) {
        return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + byte_ofs);
    };
    unsigned nb[4];  // byte offset of the lane's four consecutive columns of fragment nf
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
        if (GEGLU) nb[nf] = 4u * (unsigned)min(n_base + hv_perm_row_geglu(nf, 4 * quad), p.N - 4);
        else nb[nf] = 4u * (unsigned)(min(n_base + 32 * (nf >> 1) + 8 * quad, p.N - 8) + 4 * (nf & 1));
    }
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
        c.add[nf] = p.bias != nullptr ? ld4(p.bias, nb[nf]) : f32x4{0.f, 0.f, 0.f, 0.f};
        c.cs[nf] = LN ? ld4(p.colsum, nb[nf]) : f32x4{0.f, 0.f, 0.f, 0.f};
        c.tab[nf] = tab_row != nullptr ? ld4(tab_row, nb[nf]) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
// (after the loads have landed) add += table row: the order of hv_gemm.h's epilogues -- bias first, then the table
HV_DEV void hv_gemm4_fold_cols(HvGemm4Cols& c, bool have_tab) {
    if (have_tab) {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) c.add[nf] += c.tab[nf];
    }
}

template <int NMF>
HV_DEV void hv_gemm4_pack_ln(f32x4 (&acc)[4][NMF], const HvGemm4Cols& c, const float (&mean)[NMF], const float (&rstd)[NMF],
                             u32x4 (&outp)[NMF][2]) {
    constexpr int G = HV_GEMM_EPI_G;
#pragma unroll
    for (int g = 0; g < NMF; g += G) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int mf = g + j;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4 o;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int nf = 2 * h + k;
                    f32x4 v = hv_acc_take(acc[nf][mf]);
                    v = rstd[mf] * (v - mean[mf] * c.cs[nf]);
                    v += c.add[nf];
                    o[2 * k] = hv_pack2(v[0], v[1]);
                    o[2 * k + 1] = hv_pack2(v[2], v[3]);
                }
                outp[mf][h] = o;
            }
        }
#if !defined(HV_EMU)
        __builtin_amdgcn_sched_barrier(0);  // the next group's accumulator reads are not hoisted over it (register budget)
#endif
    }
}

// bias (+ table row) + residual, permuted channels: the arithmetic of hv_gemm_epilogue_fast_perm<NMF, false, true> (hv_gemm.h).
// res[mf][h]: the residual's 8 bf16 at the lane's row of fragment mf, channels 32 h + 8 quad .. (loaded by the caller, one group
// of loads for the whole 64-channel block)
template <int NMF>
HV_DEV void hv_gemm4_pack_res(f32x4 (&acc)[4][NMF], const HvGemm4Cols& c, const u32x4 (&res)[NMF][2], u32x4 (&outp)[NMF][2]) {
    constexpr int G = HV_GEMM_EPI_G;
#pragma unroll
    for (int g = 0; g < NMF; g += G) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int mf = g + j;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4 o;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int nf = 2 * h + k;
                    f32x4 v = hv_acc_take(acc[nf][mf]);
                    v += c.add[nf];
                    const unsigned r0 = res[mf][h][2 * k], r1 = res[mf][h][2 * k + 1];
                    v += f32x4{hv_bf2f((bf16_t)(r0 & 0xffff)), hv_bf2f((bf16_t)(r0 >> 16)), hv_bf2f((bf16_t)(r1 & 0xffff)),
                               hv_bf2f((bf16_t)(r1 >> 16))};
                    o[2 * k] = hv_pack2(v[0], v[1]);
                    o[2 * k + 1] = hv_pack2(v[2], v[3]);
                }
                outp[mf][h] = o;
            }
        }
#if !defined(HV_EMU)
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
}

template <int NMF>
HV_DEV void hv_gemm4_pack_geglu(f32x4 (&acc)[4][NMF], const HvGemm4Cols& c, const float (&mean)[NMF], const float (&rstd)[NMF],
                                u32x4 (&outp)[NMF]) {
    constexpr int G = HV_GEMM_EPI_G;
#pragma unroll
    for (int g = 0; g < NMF; g += G) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int mf = g + j;
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x4 h = rstd[mf] * (hv_acc_take(acc[2 * k][mf]) - mean[mf] * c.cs[2 * k]) + c.add[2 * k];
                const f32x4 gt = rstd[mf] * (hv_acc_take(acc[2 * k + 1][mf]) - mean[mf] * c.cs[2 * k + 1]) + c.add[2 * k + 1];
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
}

// NMF: row fragments per wave = 8 (256 x 256 x 64 tile, waves of 128 x 128) or 6 (192 x 256 x 64 tile, waves of 96 x 128:
//        192 accumulator registers -- the form that has room for the deferred results; M % 192 == 0 holds at every level of
//        the denoising path, and a 96-row wave block is one positional-encoding period even at level 3).
// DEFER: 0 = the epilogues of hv_gemm.h, stores issued at once (every output form);
//        1 = LayerNorm-fold form, permuted channels: 4 NMF packed 16-byte results per wave and tile, stored NMF per k-tile
//            behind the first four k-tiles of the next tile;  2 = LayerNorm fold + GEGLU: 2 NMF results, NMF / 2 per k-tile.
//            (K >= 320: five k-tiles)   3 = bias (+ table row) + residual, in place or not (N % 256 == 0: no tile overlap), as 1.
#ifdef HV_W4_TRACE
// timing build (tools/build_variant.sh w4trace k_gemm -DHV_W4_TRACE): per workgroup, wave 0 accumulates s_memtime ticks spent
// [0] in the whole kernel, [1] in the counted vmcnt waits, [2] at the barriers, [3] in the epilogue (pack), [4] k-tiles, [5] tiles
__device__ unsigned long long g_hv_w4_trace[256 * 8];
#endif

template <bool PERM, int DEFER, int NMF>
__global__ __launch_bounds__(256, 1) void hv_gemm_w4_kernel(HvGemmParams p, int gm, int form) {
    static_assert(DEFER == 0 || PERM, "the deferred forms use the permuted channel assignment");
    static_assert(NMF == 8 || NMF == 6, "wave sub-tiles of 128 or 96 rows");
    constexpr int BM = 32 * NMF, BN = 256, BK = 64, XS = 3, WS = 2;
    constexpr int WTM = 16 * NMF;  // rows of a wave's sub-tile
    constexpr int XT = BM * BK * 2, WT = BN * BK * 2;
    constexpr int NB = 2 * NMF;    // blocks of 8 MFMAs per k-tile: (k half, X fragment)
    constexpr int NP = 8 + NMF;    // LDS-DMA pieces per wave and round: 8 W + NMF X
    __shared__ __attribute__((aligned(16))) unsigned char smem[XS * XT + WS * WT];
    unsigned char* const xring = smem;
    unsigned char* const wring = smem + XS * XT;

    const int tid = threadIdx.x, lane = tid & 63;
#ifndef HV_EMU
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#else
    const int wave = tid >> 6;
#endif
    const int wm = wave & 1, wn = wave >> 1;
    const int r16 = lane & 15, quad = lane >> 4;

    // ---- tile raster and persistent walk.
    // gm >= 0: as hv_gemm_glds_kernel (XCD x owns a contiguous range of tile indices, its workgroups take every tstep-th;
    //          gm > 1: gm m-blocks per n-step).
    // gm < 0:  UNIT raster, U = -gm column tiles per unit (U divides tiles_n; chosen by hv_gemm_launch).  A unit is U
    //          consecutive column tiles of ONE row block; XCD x owns a contiguous range of units, workgroup w takes every
    //          tstep-th unit and walks its column tiles one after the other.  Why: with the tile raster the tiles_n workgroups
    //          that share an X row block run it AT THE SAME TIME, so every one of them sees the HBM latency of every X piece
    //          (the L2 merges the misses but nobody hits) -- and a CU can keep only so many missed lines in flight: the trace
    //          build showed the level-0 QKV k-tile (4 sharers, all in step) at 4 200 cycles against 2 150 for an L2-resident X
    //          (profiles/r06_s4_w4_trace.txt).  With units a workgroup's first tile of a row block pulls X into the XCD's L2
    //          (K <= 640: 123 / 246 KB per workgroup) and its next U - 1 tiles hit; all workgroups of the XCD walk the same
    //          column tiles, so W stays hot as before.
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = p.M / BM;  // M % BM == 0 (hv_gemm_choose)
    const int U = gm < 0 ? -gm : 1;
    const int units_per_rb = tiles_n / U;  // (U divides tiles_n)
    const int total = gm < 0 ? tiles_m * units_per_rb : tiles_n * tiles_m;  // units resp. tiles
    // Deferred forms: a ragged last column tile is moved LEFT onto the previous one (n0 = N - 256: N = 960 -> tiles at 0,
    // 256, 512, 704) instead of being clamped: every column of every tile exists, so every deferred store is unconditional;
    // the 64 .. 192 columns computed twice get the same bits from both workgroups (no residual: Y is write-only in these forms).
    // ti: index in the workgroup's own sequence space -- tile index (gm >= 0) or unit index * U + column tile inside the unit.
    auto tile_origin = [&](int ti, int& m0, int& n0) __attribute__((always_inline)) {
        if (gm < 0) {
            const int unit = ti / U, cu = ti - unit * U;
            const int rb = unit / units_per_rb, ub = unit - rb * units_per_rb;
            m0 = rb * BM;
            n0 = (ub * U + cu) * BN;
        } else if (gm <= 1) {
            m0 = (ti / tiles_n) * BM;
            n0 = (ti % tiles_n) * BN;
        } else {
            const int per_group = gm * tiles_n;
            const int g = ti / per_group, r = ti - g * per_group;
            const int rows = max(1, min(gm, tiles_m - g * gm));
            m0 = (g * gm + r % rows) * BM;
            n0 = (r / rows) * BN;
        }
        if constexpr (DEFER != 0) n0 = min(n0, p.N - BN);
    };
    const int wg_per_xcd = gridDim.x / 8;
    const int xcd = blockIdx.x % 8, wg = blockIdx.x / 8;
    const int per_xcd = (total + 7) / 8;
    const int t_begin = xcd * per_xcd;
    const int t_end = min(total, t_begin + per_xcd);
    if (t_begin + wg >= t_end) return;
    const int my_units = (t_end - (t_begin + wg) + wg_per_xcd - 1) / wg_per_xcd;  // tiles (gm >= 0) or units
    // the workgroup's sequence: position j = 0 .. my_tiles - 1 -> ti (see tile_origin)
    const int my_tiles = my_units * U;
    const int first = (t_begin + wg) * U;
    const int tstep = wg_per_xcd;  // in units
    auto seq_next = [&](int ti) __attribute__((always_inline)) {  // the sequence position after ti
        if (U == 1) return ti + tstep;
        return (ti + 1) % U != 0 ? ti + 1 : ti + 1 + (tstep - 1) * U;
    };
    const int last_tile = (t_begin + wg + (my_units - 1) * tstep) * U + (U - 1);
    const int nk = p.K / BK;

    // ---- LDS-DMA streams.  Piece q of a wave covers tile rows 8 (wave + 4 q) + lane / 8 (1 KiB: 8 rows x 128 B); both
    // swizzles look at row bits 0..3 only, which do not depend on q: ONE per-lane byte offset per operand, everything that
    // moves (q, k-tile, tile) is in the wave-uniform base.
    const int sub = lane >> 3, slot8 = lane & 7;
    const int trow0 = 8 * wave + sub;
    const unsigned xlane = ((unsigned)sub * (unsigned)p.ldx + (unsigned)((slot8 ^ ((trow0 >> 1) & 7)) * 8)) * 2u;
    const unsigned wlane = ((unsigned)sub * (unsigned)p.K + (unsigned)((slot8 ^ (PERM ? hv_wperm_swizzle(trow0) : ((trow0 >> 1) & 7))) * 8)) * 2u;
    const long xq_stride = 32L * p.ldx * 2, wq_stride = 32L * p.K * 2;  // 32 rows per piece index
    // X stream (two k-tiles ahead of the consumer on the 3-slot ring), W stream (one ahead on the 2-slot ring).  Past the
    // workgroup's last k-tile the streams keep re-issuing its last tile (valid addresses, slots nobody reads): the counted
    // waits stay exact without a tail case.
    int ix_tile = first, ix_k = 0, ix_slot = 0;
    int iw_tile = first, iw_k = 0, iw_slot = 0;
    const char* xsrc;  // piece 0 of the X k-tile being issued
    const char* wsrc;
    int w_qmax;  // last piece index whose rows lie inside N (ragged last column tile: N % 64 == 0, pieces are 8 rows)
    auto set_x_tile = [&]() __attribute__((always_inline)) {
        int m0, n0;
        tile_origin(ix_tile, m0, n0);
        xsrc = reinterpret_cast<const char*>(p.X) + ((long)(m0 + 8 * wave) * p.ldx) * 2;
    };
    auto set_w_tile = [&]() __attribute__((always_inline)) {
        int m0, n0;
        tile_origin(iw_tile, m0, n0);
        wsrc = reinterpret_cast<const char*>(p.W) + ((long)(n0 + 8 * wave) * p.K) * 2;
        w_qmax = DEFER != 0 ? 7 : min(7, (p.N - 8 - n0 - 8 * wave) / 32);
    };
    set_x_tile();
    set_w_tile();
    auto issue_x = [&](int q) __attribute__((always_inline)) {
        hv_glds16_u(xsrc + q * xq_stride, xlane, xring + ix_slot * XT + (wave + 4 * q) * 1024);
    };
    auto issue_w = [&](int q) __attribute__((always_inline)) {
        hv_glds16_u(wsrc + min(q, w_qmax) * wq_stride, wlane, wring + iw_slot * WT + (wave + 4 * q) * 1024);
    };
    auto advance_x = [&]() __attribute__((always_inline)) {
        if (++ix_slot == XS) ix_slot = 0;
        xsrc += BK * 2;
        if (++ix_k == nk) {
            ix_k = 0;
            if (ix_tile != last_tile) ix_tile = seq_next(ix_tile);
            set_x_tile();
        }
    };
    auto advance_w = [&]() __attribute__((always_inline)) {
        if (++iw_slot == WS) iw_slot = 0;
        wsrc += BK * 2;
        if (++iw_k == nk) {
            iw_k = 0;
            if (iw_tile != last_tile) iw_tile = seq_next(iw_tile);
            set_w_tile();
        }
    };
    // piece j of a round: 0..7 = the W pieces q = j of the next k-tile, 8..NP-1 = the X pieces q = j - 8 of the one after it
    auto issue_piece = [&](auto J) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        if constexpr (j < 8) {
            issue_w(j);
            if constexpr (j == 7) advance_w();
        } else {
            issue_x(j - 8);
            if constexpr (j == NP - 1) advance_x();
        }
    };

    // ---- fragments.  The wave multiplies X rows [WTM wm, +WTM) with W rows [128 wn, +128): NMF x 8 fragments of 16 x 16.
    // W fragment nf = 4 h + f is row block h (64 channels: the unit of the epilogues) under the channel assignment of the form.
    // Every fragment address is ONE per-lane byte offset per operand and k half + a compile-time constant (the ds_read's
    // offset field): X rows 16 apart (2048 B; the row swizzle looks at row bits 1..3 only); W rows under either permuted
    // assignment differ from fragment 0's by 4 / 32 / 36 (plain: hv_perm_row) or 16 / 4 / 20 (GEGLU: hv_perm_row_geglu)
    // rows + 64 h -- bits that neither carry into nor belong to the swizzle's (row bits 0, 1, 3).
    const bool geglu_rows = PERM && form == HV_FORM_LN_GEGLU;
    const int wrow0 = 128 * wn + (!PERM ? r16 : (geglu_rows ? hv_perm_row_geglu(0, r16) : hv_perm_row(0, r16)));
    unsigned wl[2], xl[2];  // lane offsets inside a slot, per k half
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        wl[kk] = (unsigned)(PERM ? hv_swz_wperm(wrow0, kk * 4 + quad) : hv_swz<BK>(wrow0, kk * 4 + quad));
        xl[kk] = (unsigned)hv_swz<BK>(WTM * wm + r16, kk * 4 + quad);
    }
    // byte distance of W fragment f (0..3) from fragment 0 inside a 64-row block: rows x 128 B
    const unsigned wf1 = geglu_rows ? 16u * 128u : (PERM ? 4u * 128u : 16u * 128u);   // f = 1
    const unsigned wf2 = geglu_rows ? 4u * 128u : (PERM ? 32u * 128u : 32u * 128u);   // f = 2   (f = 3: wf1 + wf2)
    auto rd_w = [&](const unsigned char* ws, int kk, int nf) __attribute__((always_inline)) {
        const int f = nf & 3;
        const unsigned d = (unsigned)(nf >> 2) * 8192u + ((f & 1) ? wf1 : 0u) + ((f & 2) ? wf2 : 0u);
        return hv_as_bf16x8(hv_ld16(ws + wl[kk] + d));
    };
    auto rd_x = [&](const unsigned char* xs, int kk, int mf) __attribute__((always_inline)) {
        return hv_as_bf16x8(hv_ld16(xs + xl[kk] + (unsigned)mf * 2048u));
    };

    f32x4 acc[8][NMF];  // [nf][mf]
    bf16x8 wf[8], xr[4];  // ONE set of W fragments: the next k half's replace them in place (blocks NMF - 1 and NB - 1)
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto fence = [&]() __attribute__((always_inline)) {
#ifndef HV_EMU
        __builtin_amdgcn_sched_barrier(0);
#endif
    };
    // block b of a k-tile: kk = b / NMF, X fragment mf = b % NMF (in xr[b % 4]) against the eight W fragments of kk.
    // RELOAD (blocks NMF - 1 and NB - 1): W fragment nf of the NEXT k half (wnext: its slot, knext: its k half) is read into
    // wf[nf] right behind the MFMA that used wf[nf] last -- eight MFMAs (128 matrix cycles) ahead of its first use.
    auto mfma_block = [&](auto B, auto FIRST, auto RELOAD, const unsigned char* wnext, int knext) __attribute__((always_inline)) {
        constexpr int b = decltype(B)::value, kk = b / NMF, mf = b % NMF;
        constexpr bool zero_c = decltype(FIRST)::value != 0 && kk == 0;
#pragma unroll
        for (int nf = 0; nf < 8; ++nf) {
            acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nf], xr[b % 4], zero_c ? zero4 : acc[nf][mf], 0, 0, 0);
            if constexpr (decltype(RELOAD)::value != 0) wf[nf] = rd_w(wnext, knext, nf);
        }
    };

    // ---- prologue: W k-tile 0 and X k-tiles 0, 1 of the stream; everything landed before the first barrier
#pragma unroll
    for (int q = 0; q < 8; ++q) issue_w(q);
    advance_w();
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int q = 0; q < NMF; ++q) issue_x(q);
        advance_x();
    }
    int cx_slot = 0, cw_slot = 0;  // consumer slots of the k-tile whose main blocks run next
    int c_tile = first;
    hv_vm_wait<0>();

    // One k-tile of the flattened (tile, k) sequence, rotated by two blocks:
    //   barrier s  ->  TAIL blocks NB - 2, NB - 1 of k-tile s - 1 (their fragments are in registers; they carry the first
    //   fragment reads of k-tile s and pieces 0, 1 of the round)  ->  the epilogue, when k-tile s - 1 ended a tile  ->  MAIN
    //   blocks 0 .. NB - 3 of k-tile s with pieces 2 .. NP - 1 (NMF = 6: twelve pieces in ten blocks, the first two carry two).
    // Slots: barrier s waits lgkmcnt(0) first, so behind it every wave has finished ALL fragment reads of k-tile s - 1 (the
    //   reads of its tail blocks were issued two blocks earlier) and the round may overwrite that k-tile's slots: pieces
    //   0..7 = W(s + 1) on the 2-slot W ring, pieces 8.. = X(s + 2) on the 3-slot X ring.
    // Waits: barrier s needs W(s) and X(s).  W(s) is pieces 0..7 of the round behind barrier s - 1, X(s) is older (the round
    //   before that, or the prologue); the only younger requests are the NMF X(s + 1) pieces and the NS deferred stores
    //   that k-tile s - 1 issued behind them: vmcnt(NMF + NS), or vmcnt(NMF) when it had none to issue.  A store is
    //   therefore waited for one whole k-tile after its own; W(s + 1) has at least eight blocks (~1100 cycles, L2 hits)
    //   between its last piece and the wait, X(s + 2) a whole k-tile more.
    // Deferred stores (DEFER != 0): the tile's results are packed at its end (outp).  A quarter is stored at once; the rest
    //   goes out behind the first FIVE k-tiles of the NEXT tile, 3 - 4 (GEGLU: 1 - 2) stores per k-tile, each behind the X piece
    //   of a block (blocks SB ..).  Why so thin: all 256 workgroups run in step, and the chip writes ~5 TB/s = ~10 B per
    //   cycle and CU; a wave whose store finds the write path busy stalls IN the k-tile (in-order issue: its MFMAs wait too) --
    //   with a group of 6 stores per k-tile the trace build showed the k-tiles that carry stores at 4 050 cycles against
    //   2 150 without (profiles/r06_s4_w4_trace.txt).
    constexpr int EXTRA = NP - NB;                  // main blocks that carry two pieces (0 or 2)
    constexpr int SB = 6 - EXTRA;                   // first main block whose pieces are all X pieces: 6 (NMF 8) / 4 (NMF 6)
    static_assert(NB - 2 - SB == NMF, "one store slot per X piece block");
    constexpr int NOUT = (DEFER == 1 || DEFER == 3) ? 4 * NMF : (DEFER == 2 ? 2 * NMF : 1);
    constexpr int NS = DEFER != 0 ? NOUT / 4 : 0;   // group 0: stored at once by the epilogue
    constexpr int ND = DEFER != 0 ? NOUT - NS : 0;  // deferred: spread over the first FIVE k-tiles of the next tile
    u32x4 outp[NOUT];  // DEFER 1: [h][mf][c] (c: channel halves 32 c + 8 quad of the block), DEFER 2: [h][mf]
    // the wave's sub-tile origin in Y of the tile whose results are pending.  Before the first tile's own epilogue it points
    // at the FIRST tile: the store slots of that tile's k-tiles are unconditional (a conditional store splits the k-tile into
    // a dozen basic blocks: 120 spilled registers) and write whatever outp holds there -- overwritten by the same lanes' real
    // results a tile later (same wave, same addresses, program order)
    const char* ydef = nullptr;
    if constexpr (DEFER != 0) {
        int m0, n0;
        tile_origin(first, m0, n0);
        const int mb = m0 + WTM * wm, nb = n0 + 128 * wn;
        ydef = reinterpret_cast<const char*>(p.Y) + ((long)mb * p.ldy + (DEFER != 2 ? nb : (nb >> 1))) * 2;
    }
    const unsigned ylane = ((unsigned)r16 * (unsigned)p.ldy + 8u * (unsigned)quad) * 2u;
    const unsigned yfrag = 16u * (unsigned)p.ldy * 2u;  // bytes between row fragments
    auto store_out = [&](int i) __attribute__((always_inline)) {  // (i is a constant after inlining)
        if constexpr (DEFER == 1 || DEFER == 3) {
            const int h = i / (2 * NMF), mf = (i / 2) % NMF, c = i & 1;
            hv_st16(const_cast<char*>(ydef) + ((unsigned)mf * yfrag + (unsigned)(128 * h + 64 * c)) + ylane, outp[i]);
        } else if constexpr (DEFER == 2) {
            const int h = i / NMF, mf = i % NMF;
            hv_st16(const_cast<char*>(ydef) + ((unsigned)mf * yfrag + (unsigned)(64 * h)) + ylane, outp[i]);
        }
    };
    if constexpr (DEFER == 3) {
        // the residual form may store IN PLACE (Y is the residual): the first tile's unconditional store slots must not put
        // garbage where that tile's own epilogue reads the residual later -- they write back what is there
        hv_static_for<NOUT>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value, h = i / (2 * NMF), mf = (i / 2) % NMF, c = i & 1;
            outp[i] = hv_ld16(ydef + ((unsigned)mf * yfrag + (unsigned)(128 * h + 64 * c)) + ylane);
        });
    }
    auto pack_tile = [&](int ti) __attribute__((always_inline)) {
        int m0, n0;
        tile_origin(ti, m0, n0);
        const int mb = m0 + WTM * wm, nb = n0 + 128 * wn;
        if constexpr (DEFER == 0) {
            hv_gemm_epilogue_form<NMF, PERM, 0>(form, p, reinterpret_cast<f32x4(&)[4][NMF]>(acc[0]), mb, nb, r16, quad);
            hv_gemm_epilogue_form<NMF, PERM, 0>(form, p, reinterpret_cast<f32x4(&)[4][NMF]>(acc[4]), mb, nb + 64, r16, quad);
        } else if constexpr (DEFER == 3) {
            const float* tab = nullptr;
            if (p.pe != nullptr) tab = p.pe + (long)((mb / p.pe_period) % p.pe_frames) * p.N;
            else if (p.rowvec != nullptr) tab = p.rowvec + (long)(mb / p.rowvec_period) * p.N;
            hv_acc_settle();
#pragma unroll
            for (int h = 0; h < 2; ++h) {  // one 64-channel block at a time: its residual rows are 12 loads in flight (48 registers)
                HvGemm4Cols c0;
                hv_gemm4_load_cols<false, false>(p, nb + 64 * h, quad, tab, c0);
                u32x4 res[NMF][2];
#pragma unroll
                for (int mf = 0; mf < NMF; ++mf)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        res[mf][c] = hv_ld16(reinterpret_cast<const char*>(p.residual) +
                                             (((unsigned)(mb + 16 * mf + r16) * (unsigned)p.ldr + (unsigned)(nb + 64 * h + 32 * c + 8 * quad)) * 2u));
                fence();
                hv_gemm4_fold_cols(c0, tab != nullptr);
                hv_gemm4_pack_res<NMF>(reinterpret_cast<f32x4(&)[4][NMF]>(acc[4 * h]), c0, res, reinterpret_cast<u32x4(&)[NMF][2]>(outp[2 * NMF * h]));
                fence();
            }
            ydef = reinterpret_cast<const char*>(p.Y) + ((long)mb * p.ldy + nb) * 2;
#pragma unroll
            for (int i = 0; i < NS; ++i) store_out(i);
        } else {
            const float* tab = nullptr;  // one table row per wave sub-tile (hv_gemm_fast_form(p, WTM))
            if (p.pe != nullptr) tab = p.pe + (long)((mb / p.pe_period) % p.pe_frames) * p.N;
            else if (p.rowvec != nullptr) tab = p.rowvec + (long)(mb / p.rowvec_period) * p.N;
            // phase 1: every operand of the tile's epilogue requested back to back (one round trip)
            HvGemm4Cols c0, c1;
            float mean[NMF], rstd[NMF];
            hv_gemm4_load_cols<DEFER == 2>(p, nb, quad, tab, c0);
            hv_gemm4_load_cols<DEFER == 2>(p, nb + 64, quad, tab, c1);
#pragma unroll
            for (int mf = 0; mf < NMF; ++mf) {
                const unsigned mo = 4u * (unsigned)(mb + 16 * mf + r16);  // (M % BM == 0: every row exists)
                mean[mf] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.row_mean) + mo);
                rstd[mf] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.row_rstd) + mo);
            }
            fence();
            hv_gemm4_fold_cols(c0, tab != nullptr);
            hv_gemm4_fold_cols(c1, tab != nullptr);
            hv_acc_settle();
            // phase 2
            if constexpr (DEFER == 1) {
                hv_gemm4_pack_ln<NMF>(reinterpret_cast<f32x4(&)[4][NMF]>(acc[0]), c0, mean, rstd, reinterpret_cast<u32x4(&)[NMF][2]>(outp[0]));
                hv_gemm4_pack_ln<NMF>(reinterpret_cast<f32x4(&)[4][NMF]>(acc[4]), c1, mean, rstd, reinterpret_cast<u32x4(&)[NMF][2]>(outp[2 * NMF]));
                ydef = reinterpret_cast<const char*>(p.Y) + ((long)mb * p.ldy + nb) * 2;
            } else {
                hv_gemm4_pack_geglu<NMF>(reinterpret_cast<f32x4(&)[4][NMF]>(acc[0]), c0, mean, rstd, reinterpret_cast<u32x4(&)[NMF]>(outp[0]));
                hv_gemm4_pack_geglu<NMF>(reinterpret_cast<f32x4(&)[4][NMF]>(acc[4]), c1, mean, rstd, reinterpret_cast<u32x4(&)[NMF]>(outp[NMF]));
                ydef = reinterpret_cast<const char*>(p.Y) + ((long)mb * p.ldy + (nb >> 1)) * 2;
            }
            // group 0 goes out at once (a quarter of the tile: NS stores): the k-tile that follows then holds three groups
            // beside its fragments, not four (the register file is full at four: the packed results were spilled to scratch,
            // and every reload waits vmcnt(0) in the middle of a k-tile)
#pragma unroll
            for (int i = 0; i < NS; ++i) store_out(i);
        }
    };
#ifdef HV_W4_TRACE
    unsigned long long tr_wait = 0, tr_bar = 0, tr_pack = 0, tr_steps = 0, tr_tiles = 0;
    const unsigned long long tr_begin = __builtin_amdgcn_s_memtime();
#endif
    // V: k-tile of the tile (0..4: it issues its share of the deferred stores; -1: a later one, none; 0 is also the FIRST
    // k-tile of a tile: it carries the previous tile's epilogue and starts the accumulators from zero).  PH: the number of
    // stores the previous k-tile issued (the wait leaves that many more requests in flight).
    auto round_step = [&](auto VV, auto PHV, bool have_tail) __attribute__((always_inline)) {
        constexpr int V = decltype(VV)::value, PH = decltype(PHV)::value;
        constexpr bool FIRST = V == 0;
#ifdef HV_W4_TRACE
        const unsigned long long tr0 = __builtin_amdgcn_s_memtime();
#endif
        if constexpr (NS > 0 && V == 0) {
            // (the previous k-tile is the previous tile's last: with five k-tiles per tile that is the one that issued the
            //  last C4 deferred stores -- they stay in flight; with more, it issued none)
            constexpr int C4 = ND / 5 + (4 < ND % 5 ? 1 : 0);
            if (nk == 5) hv_vm_wait<NMF + C4>();
            else hv_vm_wait<NMF>();
        } else {
            hv_vm_wait<NMF + (PH > 0 ? PH : 0)>();
        }
#ifdef HV_W4_TRACE
        const unsigned long long tr1 = __builtin_amdgcn_s_memtime();
#endif
        hv_barrier_raw();
#ifdef HV_W4_TRACE
        const unsigned long long tr2 = __builtin_amdgcn_s_memtime();
        tr_wait += tr1 - tr0;
        tr_bar += tr2 - tr1;
        ++tr_steps;
#endif
        const unsigned char* xs = xring + cx_slot * XT;
        const unsigned char* ws = wring + cw_slot * WT;
        if (++cx_slot == XS) cx_slot = 0;
        if (++cw_slot == WS) cw_slot = 0;
        if constexpr (!FIRST) {
            // tail blocks of the previous k-tile with the first fragment reads of this one
            xr[0] = rd_x(xs, 0, 0);
            issue_piece(HvInt<0>{});
            mfma_block(HvInt<NB - 2>{}, HvInt<0>{}, HvInt<0>{}, nullptr, 0);
            fence();
            xr[1] = rd_x(xs, 0, 1);
            issue_piece(HvInt<1>{});
            mfma_block(HvInt<NB - 1>{}, HvInt<0>{}, HvInt<1>{}, ws, 0);
            fence();
        } else {
            // the previous tile's last two blocks, then its epilogue; the first fragments of this k-tile are read behind it
            // (registers: the packed results and the epilogue's operands are live in between)
            issue_piece(HvInt<0>{});
            if (have_tail) mfma_block(HvInt<NB - 2>{}, HvInt<0>{}, HvInt<0>{}, nullptr, 0);
            fence();
            issue_piece(HvInt<1>{});
            if (have_tail) mfma_block(HvInt<NB - 1>{}, HvInt<0>{}, HvInt<0>{}, nullptr, 0);
            fence();
            if (have_tail) {
#ifdef HV_W4_TRACE
                const unsigned long long tp0 = __builtin_amdgcn_s_memtime();
#endif
                pack_tile(c_tile);
                c_tile = seq_next(c_tile);
#ifdef HV_W4_TRACE
                tr_pack += __builtin_amdgcn_s_memtime() - tp0;
                ++tr_tiles;
#endif
            }
            fence();
#pragma unroll
            for (int nf = 0; nf < 8; ++nf) wf[nf] = rd_w(ws, 0, nf);
            xr[0] = rd_x(xs, 0, 0);
            xr[1] = rd_x(xs, 0, 1);
            fence();
        }
        hv_static_for<NB - 2>([&](auto B) __attribute__((always_inline)) {
            constexpr int b = decltype(B)::value;
            constexpr int j0 = 2 + b + (b < EXTRA ? b : EXTRA);  // first piece of this block
            xr[(b + 2) % 4] = rd_x(xs, (b + 2) / NMF, (b + 2) % NMF);
            issue_piece(HvInt<j0>{});
            if constexpr (b < EXTRA) issue_piece(HvInt<j0 + 1>{});
            mfma_block(B, HvInt<FIRST ? 1 : 0>{}, HvInt<(b == NMF - 1) ? 1 : 0>{}, ws, 1);
            if constexpr (NS > 0 && V >= 0 && b >= SB) {
                // deferred stores of this k-tile: cnt of them on the NMF blocks that carry X pieces (younger than W pieces
                // 0..7 of this round: see Waits), evenly spaced
                constexpr int cnt = ND / 5 + (V < ND % 5 ? 1 : 0);
                constexpr int start = NS + V * (ND / 5) + (V < ND % 5 ? V : ND % 5);
                hv_static_for<cnt>([&](auto I) __attribute__((always_inline)) {
                    constexpr int i = decltype(I)::value;
                    if constexpr (SB + (i * NMF) / cnt == b) store_out(start + i);
                });
            }
            fence();
        });
    };
    // (the first wait is trivially satisfied: the prologue's requests have landed)
    for (int t = 0; t < my_tiles; ++t) {
        round_step(HvInt<0>{}, HvInt<-1>{}, t > 0);
        if constexpr (NS > 0) {  // (nk >= 5: hv_gemm_launch)
            constexpr int C0 = ND / 5 + (0 < ND % 5 ? 1 : 0), C1 = ND / 5 + (1 < ND % 5 ? 1 : 0), C2 = ND / 5 + (2 < ND % 5 ? 1 : 0),
                          C3 = ND / 5 + (3 < ND % 5 ? 1 : 0), C4 = ND / 5 + (4 < ND % 5 ? 1 : 0);
            round_step(HvInt<1>{}, HvInt<C0>{}, true);
            round_step(HvInt<2>{}, HvInt<C1>{}, true);
            round_step(HvInt<3>{}, HvInt<C2>{}, true);
            round_step(HvInt<4>{}, HvInt<C3>{}, true);
            if (nk > 5) round_step(HvInt<-1>{}, HvInt<C4>{}, true);
            for (int k = 6; k < nk; ++k) round_step(HvInt<-1>{}, HvInt<-1>{}, true);
        } else {
            for (int k = 1; k < nk; ++k) round_step(HvInt<-1>{}, HvInt<-1>{}, true);
        }
    }
    // the last k-tile's tail blocks, the last tile's epilogue, and its results stored at once
    mfma_block(HvInt<NB - 2>{}, HvInt<0>{}, HvInt<0>{}, nullptr, 0);
    mfma_block(HvInt<NB - 1>{}, HvInt<0>{}, HvInt<0>{}, nullptr, 0);
    pack_tile(c_tile);
    if constexpr (NS > 0) {
        hv_static_for<NOUT - NS>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = NS + decltype(I)::value;
            store_out(i);
        });
    }
    hv_vm_wait<0>();  // (the streams' surplus pieces)
#ifdef HV_W4_TRACE
    if (tid == 0 && blockIdx.x < 256) {
        unsigned long long* t = g_hv_w4_trace + blockIdx.x * 8;
        t[0] = __builtin_amdgcn_s_memtime() - tr_begin;
        t[1] = tr_wait, t[2] = tr_bar, t[3] = tr_pack, t[4] = tr_steps, t[5] = tr_tiles;
    }
#endif
}
