// hv_gemm_xs.h -- "X-stationary" GEMM for the level-0 LayerNorm-fold projections (K = 320: five k-tiles), round 6.
// Included by hv_gemm.h behind hv_gemm4.h (it uses that file's helpers).  Reference call sites: the fused QKV projection of
// the motion modules (src/models/motion_module.py:233-256) and the GEGLU input projection of every level-0 feed-forward
// (diffusers FeedForward, src/models/attention.py:427) -- M = 294 912 rows, K = 320, N = 960 / 2560.
//
// What the trace build of hv_gemm_w4_kernel showed at these shapes (profiles/r06_s4_w4_trace.txt, r06_s5_w4_units.txt): a tile
// has only five k-tiles, so (1) its epilogue -- 6 600 cycles for the LayerNorm fold, 8 700 with GEGLU, VALU work with the matrix
// pipe idle -- is a third of the tile, and (2) the k-tile itself runs at 2 700 - 3 250 cycles against 2 150 for an L2-resident
// X: every k-tile streams 24 KiB of X beside 32 KiB of W.  Both have the same cure at K = 320: a 192-row block of X is
// 120 KiB -- it FITS in LDS.  So:
//   * X-stationary: the workgroup keeps its row block's five X k-tiles resident (120 KiB) and walks the U column tiles of a
//     unit (hv_gemm_w4_kernel's unit raster) over it; only W streams (128-column tiles: 16 KiB per k-tile on a 2-slot ring,
//     4 LDS-DMA pieces per wave and k-tile instead of 14).  The NEXT unit's X replaces the resident one slot by slot during the
//     unit's last tile, each k-tile behind the barrier that follows its last use -- four k-tiles ahead of its first.
//   * software-pipelined epilogue: a 192 x 128 tile is 96 accumulator registers per wave (2 x 2 waves of 96 x 64), so a
//     finished tile's accumulators are copied aside (96 v-registers) and its epilogue -- LayerNorm fold, GEGLU, packing,
//     stores -- runs in slices INSIDE the next tile's k-tiles 1..4, VALU instructions in the issue slots between that tile's
//     MFMAs; its operands (bias / column sums / table row / row statistics) are requested at the top of k-tile 0 and first
//     touched a k-tile later.  The stores go out one per slice: no burst, and the counted waits know how many are in flight.
// One raw barrier per k-tile, at its top (the epilogue slices fill the fragment-read latency behind it); vmcnt as in
// hv_gemm_w4_kernel: barrier s needs W(s), issued as pieces 0..3 of the previous k-tile -- what that k-tile issued behind
// them (the next unit's X pieces, the epilogue's stores) may stay in flight.
// MFMA order per accumulator is k-ascending and the epilogue arithmetic is hv_gemm4_pack_*'s: bit-identical to every other
// GEMM kernel of this library.
#pragma once

// FORM: 1 = LayerNorm fold, permuted channels (hv_perm_row);  2 = LayerNorm fold + GEGLU (hv_perm_row_geglu)
template <int FORM>
__global__ __launch_bounds__(256, 1) void hv_gemm_xs_kernel(HvGemmParams p, int U) {
    static_assert(FORM == 1 || FORM == 2, "forms");
    constexpr int BM = 192, BN = 128, BK = 64, NK = 5, NMF = 6, WS = 2;
    constexpr int XKT = BM * BK * 2, WT = BN * BK * 2;
    constexpr bool GEGLU = FORM == 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NK * XKT + WS * WT];
    unsigned char* const xres = smem;              // [k-tile][192 rows x 128 B], swizzled like every X tile of this file
    unsigned char* const wring = smem + NK * XKT;  // [slot][128 rows x 128 B], the permuted assignment's swizzle

    const int tid = threadIdx.x, lane = tid & 63;
#ifndef HV_EMU
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#else
    const int wave = tid >> 6;
#endif
    const int wm = wave & 1, wn = wave >> 1;
    const int r16 = lane & 15, quad = lane >> 4;

    // ---- unit raster (see hv_gemm_w4_kernel): a unit = U consecutive 128-column tiles of one 192-row block
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = p.M / BM;
    const int units_per_rb = tiles_n / U;
    const int total = tiles_m * units_per_rb;
    const int wg_per_xcd = gridDim.x / 8;
    const int xcd = blockIdx.x % 8, wg = blockIdx.x / 8;
    const int per_xcd = (total + 7) / 8;
    const int t_begin = xcd * per_xcd;
    const int t_end = min(total, t_begin + per_xcd);
    if (t_begin + wg >= t_end) return;
    const int my_units = (t_end - (t_begin + wg) + wg_per_xcd - 1) / wg_per_xcd;
    const int my_tiles = my_units * U;
    // tile j of this workgroup's sequence -> origin (a ragged last column tile is moved left onto its neighbour: Y is
    // write-only in these forms and the columns computed twice get the same bits)
    auto tile_origin = [&](int j, int& m0, int& n0) __attribute__((always_inline)) {
        const int unit = t_begin + wg + (j / U) * wg_per_xcd, cu = j % U;
        const int rb = unit / units_per_rb, ub = unit - rb * units_per_rb;
        m0 = rb * BM;
        n0 = min((ub * U + cu) * BN, p.N - BN);
    };

    // ---- LDS-DMA: piece q of a wave covers tile rows 8 (wave + 4 q) + lane / 8; one per-lane offset per operand
    const int sub = lane >> 3, slot8 = lane & 7;
    const int trow0 = 8 * wave + sub;
    const unsigned xlane = ((unsigned)sub * (unsigned)p.ldx + (unsigned)((slot8 ^ ((trow0 >> 1) & 7)) * 8)) * 2u;
    const unsigned wlane = ((unsigned)sub * (unsigned)p.K + (unsigned)((slot8 ^ hv_wperm_swizzle(trow0)) * 8)) * 2u;
    const long xq_stride = 32L * p.ldx * 2, wq_stride = 32L * p.K * 2;
    // W stream: k-tile (j, k) of the flattened sequence, one k-tile ahead of its use
    int iw_j = 0, iw_k = 0, iw_slot = 0;
    const char* wsrc;
    auto set_w_tile = [&]() __attribute__((always_inline)) {
        int m0, n0;
        tile_origin(min(iw_j, my_tiles - 1), m0, n0);  // (past the end: the last tile again -- pieces nobody reads)
        wsrc = reinterpret_cast<const char*>(p.W) + ((long)(n0 + 8 * wave) * p.K) * 2;
    };
    set_w_tile();
    auto issue_w = [&](int q) __attribute__((always_inline)) {
        hv_glds16_u(wsrc + q * wq_stride, wlane, wring + iw_slot * WT + (wave + 4 * q) * 1024);
    };
    auto advance_w = [&]() __attribute__((always_inline)) {
        iw_slot ^= 1;
        wsrc += BK * 2;
        if (++iw_k == NK) {
            iw_k = 0;
            ++iw_j;
            set_w_tile();
        }
    };
    // X of the row block of tile j, k-tile kt -> resident slot kt
    auto issue_x = [&](int j, int kt, int q) __attribute__((always_inline)) {
        int m0, n0;
        tile_origin(j, m0, n0);
        const char* src = reinterpret_cast<const char*>(p.X) + ((long)(m0 + 8 * wave) * p.ldx + kt * BK) * 2;
        hv_glds16_u(src + q * xq_stride, xlane, xres + kt * XKT + (wave + 4 * q) * 1024);
    };

    // ---- fragments: the wave multiplies X rows [96 wm, +96) with the 64-channel block wn of the W tile
    const int wrow0 = 64 * wn + (GEGLU ? hv_perm_row_geglu(0, r16) : hv_perm_row(0, r16));
    unsigned wl[2], xl[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        wl[kk] = (unsigned)hv_swz_wperm(wrow0, kk * 4 + quad);
        xl[kk] = (unsigned)hv_swz<BK>(96 * wm + r16, kk * 4 + quad);
    }
    constexpr unsigned wf1 = GEGLU ? 16u * 128u : 4u * 128u, wf2 = GEGLU ? 4u * 128u : 32u * 128u;
    auto rd_w = [&](const unsigned char* ws, int kk, int f) __attribute__((always_inline)) {
        return hv_as_bf16x8(hv_ld16(ws + wl[kk] + (((f & 1) ? wf1 : 0u) + ((f & 2) ? wf2 : 0u))));
    };
    auto rd_x = [&](const unsigned char* xs, int kk, int mf) __attribute__((always_inline)) {
        return hv_as_bf16x8(hv_ld16(xs + xl[kk] + (unsigned)mf * 2048u));
    };
    auto fence = [&]() __attribute__((always_inline)) {
#ifndef HV_EMU
        __builtin_amdgcn_sched_barrier(0);
#endif
    };

    f32x4 acc[4][NMF];   // this tile
    f32x4 prev[4][NMF];  // the previous tile's accumulators, copied aside at its end: what the pipelined epilogue reads
    bf16x8 wf[2][4], xr[4];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // ---- epilogue of the previous tile, in slices.  Operands: requested at the top of k-tile 0, folded at the top of k-tile 1.
    HvGemm4Cols cols;
    float mean[NMF], rstd[NMF];
    bool have_tab = false;
    const char* ybase = nullptr;  // the previous tile's sub-tile origin in Y
    const unsigned ylane = ((unsigned)r16 * (unsigned)p.ldy + 8u * (unsigned)quad) * 2u;
    const unsigned yfrag = 16u * (unsigned)p.ldy * 2u;
    auto load_operands = [&](int j) __attribute__((always_inline)) {  // for the epilogue of tile j
        int m0, n0;
        tile_origin(j, m0, n0);
        const int mb = m0 + 96 * wm, nb = n0 + 64 * wn;
        const float* tab = nullptr;  // one table row per wave sub-tile (hv_gemm_fast_form(p, 96))
        if (p.pe != nullptr) tab = p.pe + (long)((mb / p.pe_period) % p.pe_frames) * p.N;
        else if (p.rowvec != nullptr) tab = p.rowvec + (long)(mb / p.rowvec_period) * p.N;
        have_tab = tab != nullptr;
        hv_gemm4_load_cols<GEGLU>(p, nb, quad, tab, cols);
#pragma unroll
        for (int mf = 0; mf < NMF; ++mf) {
            const unsigned mo = 4u * (unsigned)(mb + 16 * mf + r16);
            mean[mf] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.row_mean) + mo);
            rstd[mf] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.row_rstd) + mo);
        }
        ybase = reinterpret_cast<const char*>(p.Y) + ((long)mb * p.ldy + (GEGLU ? (nb >> 1) : nb)) * 2;
    };
    // slice i: LayerNorm fold: 12 per tile, (row fragment i / 2, channel half i % 2) -> one 16-byte store;
    //          GEGLU: 6 per tile, row fragment i -> one 16-byte store
    constexpr int NSLICE = GEGLU ? NMF : 2 * NMF;
    auto slice = [&](int i) __attribute__((always_inline)) {  // (i is a constant after inlining)
        u32x4 o;
        if constexpr (!GEGLU) {
            const int mf = i >> 1, h = i & 1;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int nf = 2 * h + k;
                f32x4 v = prev[nf][mf];
                v = rstd[mf] * (v - mean[mf] * cols.cs[nf]);
                v += cols.add[nf];
                o[2 * k] = hv_pack2(v[0], v[1]);
                o[2 * k + 1] = hv_pack2(v[2], v[3]);
            }
            hv_st16(const_cast<char*>(ybase) + ((unsigned)mf * yfrag + (unsigned)(64 * h)) + ylane, o);
        } else {
            const int mf = i;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x4 hh = rstd[mf] * (prev[2 * k][mf] - mean[mf] * cols.cs[2 * k]) + cols.add[2 * k];
                const f32x4 gt = rstd[mf] * (prev[2 * k + 1][mf] - mean[mf] * cols.cs[2 * k + 1]) + cols.add[2 * k + 1];
                hh = hv_gelu_times(gt, hh);
                o[2 * k] = hv_pack2(hh[0], hh[1]);
                o[2 * k + 1] = hv_pack2(hh[2], hh[3]);
            }
            hv_st16(const_cast<char*>(ybase) + (unsigned)mf * yfrag + ylane, o);
        }
    };
    // slices of k-tile k (1..4): LayerNorm fold 3 per k-tile (blocks 4, 7, 10); GEGLU 2 per k-tile in k = 1..3 (blocks 4, 8)
    constexpr int SPK = GEGLU ? 2 : 3;
    auto nst = [](int k) constexpr { return (k >= 1 && (k - 1) * SPK < NSLICE) ? SPK : 0; };  // stores issued in k-tile k

    // ---- prologue: the first row block's X (all five k-tiles) and W k-tile (0, 0); everything landed before the first barrier
    for (int kt = 0; kt < NK; ++kt)
#pragma unroll
        for (int q = 0; q < NMF; ++q) issue_x(0, kt, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_w(q);
    advance_w();
    hv_vm_wait<0>();

    int cw_slot = 0;
    for (int j = 0; j < my_tiles; ++j) {
        const bool epi = j > 0;                                   // a previous tile's epilogue runs inside this tile
        const bool last_of_unit = (j % U) == U - 1 && j + 1 < my_tiles;  // the next unit's X k-tiles 0..3 stream in during this tile
        const bool first_of_unit = j > 0 && (j % U) == 0;         // ... and its k-tile 4 during k-tile 0 of its first tile
        hv_static_for<NK>([&](auto KV) __attribute__((always_inline)) {
            constexpr int k = decltype(KV)::value;
            // what the previous k-tile issued behind its W pieces may stay in flight: the next unit's X pieces, its stores
            constexpr int kp = (k + NK - 1) % NK;  // the previous k-tile's index within ITS tile
            const bool xprev = k == 0 ? false /* (j - 1, 4): slot 3 of the next unit went out there if tile j - 1 was last_of_unit */
                                      : (k == 1 ? first_of_unit : last_of_unit);
            const bool xprev0 = k == 0 && j > 0 && ((j - 1) % U) == U - 1;  // tile j - 1 was the last of its unit (and j exists)
            const bool xp = k == 0 ? xprev0 : xprev;
            const bool sp = k == 0 ? (j > 1) : epi;  // stores of the previous k-tile: (j - 1, 4) carried an epilogue iff j - 1 > 0
            constexpr int NSP = nst(kp);
            if (xp) {
                if (NSP > 0 && sp) hv_vm_wait<NMF + NSP>();
                else hv_vm_wait<NMF>();
            } else {
                if (NSP > 0 && sp) hv_vm_wait<NSP>();
                else hv_vm_wait<0>();
            }
            hv_barrier_raw();
            const unsigned char* ws = wring + cw_slot * WT;
            const unsigned char* xs = xres + k * XKT;
            cw_slot ^= 1;
            if constexpr (k == 0) {
                if (epi) load_operands(j - 1);  // older than this k-tile's W pieces: landed by the next barrier
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int f = 0; f < 4; ++f) wf[kk][f] = rd_w(ws, kk, f);
            xr[0] = rd_x(xs, 0, 0);
            xr[1] = rd_x(xs, 0, 1);
            if constexpr (k == 1) {
                if (epi) hv_gemm4_fold_cols(cols, have_tab);  // first touch of the operands: right behind the barrier
            }
            fence();
            hv_static_for<2 * NMF>([&](auto B) __attribute__((always_inline)) {
                constexpr int b = decltype(B)::value, kk = b / NMF, mf = b % NMF;
                if constexpr (b + 2 < 2 * NMF) xr[(b + 2) % 4] = rd_x(xs, (b + 2) / NMF, (b + 2) % NMF);
                if constexpr (b < 4) {
                    issue_w(b);
                    if constexpr (b == 3) advance_w();
                } else if constexpr (b < 4 + NMF) {
                    // the next unit's X k-tile into the resident slot whose last reader has passed this k-tile's barrier
                    if (k == 0 ? first_of_unit : last_of_unit) issue_x(k == 0 ? j : j + 1, (k + NK - 1) % NK, b - 4);
                }
#pragma unroll
                for (int nf = 0; nf < 4; ++nf)
                    acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][nf], xr[b % 4], (k == 0 && kk == 0) ? zero4 : acc[nf][mf], 0, 0, 0);
                if constexpr (k >= 1 && b >= 4 && (b - 4) % (GEGLU ? 4 : 3) == 0 && (b - 4) / (GEGLU ? 4 : 3) < SPK) {
                    constexpr int i = (k - 1) * SPK + (b - 4) / (GEGLU ? 4 : 3);
                    if constexpr (i < NSLICE) {
                        if (epi) slice(i);
                    }
                }
                fence();
            });
            if constexpr (k == NK - 1) {
                // the tile is complete: its accumulators aside for the pipelined epilogue (hipcc pads no hazards around the
                // accumulator reads of hv_acc_take: settle first)
                hv_acc_settle();
#pragma unroll
                for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                    for (int mf2 = 0; mf2 < NMF; ++mf2) prev[nf][mf2] = hv_acc_take(acc[nf][mf2]);
                fence();
            }
        });
    }
    // the last tile's epilogue, at once
    load_operands(my_tiles - 1);
    fence();
    hv_gemm4_fold_cols(cols, have_tab);
#pragma unroll
    for (int i = 0; i < NSLICE; ++i) slice(i);
    hv_vm_wait<0>();  // (the W stream's surplus pieces)
}
