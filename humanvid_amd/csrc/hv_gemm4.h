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

template <bool PERM>
__global__ __launch_bounds__(256, 1) void hv_gemm_w4_kernel(HvGemmParams p, int gm, int form) {
    constexpr int BM = 256, BN = 256, BK = 64, XS = 3, WS = 2;
    constexpr int XT = BM * BK * 2, WT = BN * BK * 2;
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

    // tile raster and persistent walk: as hv_gemm_glds_kernel
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = p.M / BM;  // M % 256 == 0 (hv_gemm_choose)
    const int total = tiles_n * tiles_m;
    auto tile_origin = [&](int ti, int& m0, int& n0) __attribute__((always_inline)) {
        if (gm <= 1) {
            m0 = (ti / tiles_n) * BM;
            n0 = (ti % tiles_n) * BN;
            return;
        }
        const int per_group = gm * tiles_n;
        const int g = ti / per_group, r = ti - g * per_group;
        const int rows = max(1, min(gm, tiles_m - g * gm));
        m0 = (g * gm + r % rows) * BM;
        n0 = (r / rows) * BN;
    };
    const int wg_per_xcd = gridDim.x / 8;
    const int xcd = blockIdx.x % 8, wg = blockIdx.x / 8;
    const int per_xcd = (total + 7) / 8;
    const int t_begin = xcd * per_xcd;
    const int t_end = min(total, t_begin + per_xcd);
    const int first = t_begin + wg;
    const int tstep = wg_per_xcd;
    if (first >= t_end) return;
    const int my_tiles = (t_end - first + wg_per_xcd - 1) / wg_per_xcd;
    const int last_tile = first + (my_tiles - 1) * tstep;
    const int nk = p.K / BK;

    // ---- LDS-DMA streams.  Piece q of a wave covers tile rows 8 (wave + 4 q) + lane / 8 (1 KiB: 8 rows x 128 B); both
    // swizzles look at row bits 0..3 only, which do not depend on q: ONE per-lane byte offset per operand, everything that
    // moves (q, k-tile, tile) is in the wave-uniform base.
    const int sub = lane >> 3, slot8 = lane & 7;
    const int trow0 = 8 * wave + sub;
    const unsigned xlane = ((unsigned)sub * (unsigned)p.ldx + (unsigned)((slot8 ^ ((trow0 >> 1) & 7)) * 8)) * 2u;
    const unsigned wlane = ((unsigned)sub * (unsigned)p.K + (unsigned)((slot8 ^ (PERM ? hv_wperm_swizzle(trow0) : ((trow0 >> 1) & 7))) * 8)) * 2u;
    const long xq_stride = 32L * p.ldx * 2, wq_stride = 32L * p.K * 2;  // 32 rows per piece index
    // X stream (three k-tiles ahead), W stream (two ahead).  Past the workgroup's last k-tile the streams keep re-issuing
    // its last tile (valid addresses, slots nobody reads): the counted waits stay exact without a tail case.
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
        w_qmax = min(7, (p.N - 8 - n0 - 8 * wave) / 32);
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
            if (ix_tile != last_tile) ix_tile += tstep;
            set_x_tile();
        }
    };
    auto advance_w = [&]() __attribute__((always_inline)) {
        if (++iw_slot == WS) iw_slot = 0;
        wsrc += BK * 2;
        if (++iw_k == nk) {
            iw_k = 0;
            if (iw_tile != last_tile) iw_tile += tstep;
            set_w_tile();
        }
    };
    // piece j of a round: 0..7 = W pieces q = j of the k-tile two ahead, 8..15 = X pieces q = j - 8 of the k-tile three ahead
    auto issue_piece = [&](auto J) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        if constexpr (j < 8) {
            issue_w(j);
            if constexpr (j == 7) advance_w();
        } else {
            issue_x(j - 8);
            if constexpr (j == 15) advance_x();
        }
    };

    // ---- fragments.  The wave multiplies X rows [128 wm, +128) with W rows [128 wn, +128): 8 x 8 fragments of 16 x 16.
    // W fragment nf = 4 h + f is row block h (64 channels: the unit of the epilogues) under the channel assignment of the form.
    int wrow[4];
#pragma unroll
    for (int f = 0; f < 4; ++f)
        wrow[f] = !PERM ? 16 * f + r16 : (form == HV_FORM_LN_GEGLU ? hv_perm_row_geglu(f, r16) : hv_perm_row(f, r16));
    auto rd_w = [&](const unsigned char* ws, int kk, int nf) __attribute__((always_inline)) {
        const int row = 128 * wn + 64 * (nf >> 2) + wrow[nf & 3];
        return hv_as_bf16x8(hv_ld16(ws + (PERM ? hv_swz_wperm(row, kk * 4 + quad) : hv_swz<BK>(row, kk * 4 + quad))));
    };
    auto rd_x = [&](const unsigned char* xs, int kk, int mf) __attribute__((always_inline)) {
        return hv_as_bf16x8(hv_ld16(xs + hv_swz<BK>(128 * wm + 16 * mf + r16, kk * 4 + quad)));
    };

    f32x4 acc[8][8];  // [nf][mf]
    bf16x8 wfA[8], wfB[8], xr[4];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto fence = [&]() __attribute__((always_inline)) {
#ifndef HV_EMU
        __builtin_amdgcn_sched_barrier(0);
#endif
    };
    // block b of a k-tile: kk = b / 8, X fragment mf = b % 8 (in xr[b % 4]) against the eight W fragments of kk
    auto mfma_block = [&](auto B, auto FIRST) __attribute__((always_inline)) {
        constexpr int b = decltype(B)::value, kk = b / 8, mf = b % 8;
        constexpr bool zero_c = decltype(FIRST)::value != 0 && kk == 0;
#pragma unroll
        for (int nf = 0; nf < 8; ++nf)
            acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kk ? wfB[nf] : wfA[nf], xr[b % 4], zero_c ? zero4 : acc[nf][mf], 0, 0, 0);
    };

    // ---- prologue: W k-tile 0 and X k-tiles 0, 1 of the stream; everything landed before the first barrier
#pragma unroll
    for (int q = 0; q < 8; ++q) issue_w(q);
    advance_w();
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int q = 0; q < 8; ++q) issue_x(q);
        advance_x();
    }
    int cx_slot = 0, cw_slot = 0;  // consumer slots of the k-tile whose blocks 0..13 run next
    int c_tile = first;
    hv_vm_wait<0>();

    // One k-tile of the flattened (tile, k) sequence, rotated by two blocks:
    //   barrier s  ->  tail blocks 14, 15 of k-tile s - 1 (their fragments are in registers; they carry the first fragment
    //   reads of k-tile s and pieces 0, 1 of the round)  ->  the epilogue, when k-tile s - 1 ended a tile  ->  blocks 0..13
    //   of k-tile s with pieces 2..15.
    // Slots: barrier s waits lgkmcnt(0) first, so behind it every wave has finished ALL fragment reads of k-tile s - 1 (the
    //   reads of its blocks 14, 15 were issued in blocks 12, 13) and the round may overwrite that k-tile's slots: pieces
    //   0..7 = W(s + 1) on the 2-slot W ring, pieces 8..15 = X(s + 2) on the 3-slot X ring.
    // Waits: barrier s needs W(s) and X(s).  W(s) is pieces 0..7 of the round behind barrier s - 1, X(s) is older (the round
    //   before that, or the prologue); the only younger requests are the eight X(s + 1) pieces: vmcnt(8).  W(s + 1) has at
    //   least eight blocks (~1100 cycles, L2 hits) between its last piece and the wait, X(s + 2) a whole k-tile more.
    auto round_step = [&](auto FIRSTV, bool have_tail, bool tail_ends_tile) __attribute__((always_inline)) {
        constexpr int FIRST = decltype(FIRSTV)::value;
        hv_vm_wait<8>();
        hv_barrier_raw();
        const unsigned char* xs = xring + cx_slot * XT;
        const unsigned char* ws = wring + cw_slot * WT;
        if (++cx_slot == XS) cx_slot = 0;
        if (++cw_slot == WS) cw_slot = 0;
        // tail block 14 of the previous k-tile
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) wfA[nf] = rd_w(ws, 0, nf);
        xr[0] = rd_x(xs, 0, 0);
        issue_piece(HvInt<0>{});
        if (have_tail) mfma_block(HvInt<14>{}, HvInt<0>{});
        fence();
        // tail block 15
#pragma unroll
        for (int nf = 4; nf < 8; ++nf) wfA[nf] = rd_w(ws, 0, nf);
        xr[1] = rd_x(xs, 0, 1);
        issue_piece(HvInt<1>{});
        if (have_tail) mfma_block(HvInt<15>{}, HvInt<0>{});
        fence();
        if (tail_ends_tile) {
            int m0, n0;
            tile_origin(c_tile, m0, n0);
            hv_gemm_epilogue_form<8, PERM, 0>(form, p, reinterpret_cast<f32x4(&)[4][8]>(acc[0]), m0 + 128 * wm, n0 + 128 * wn, r16, quad);
            hv_gemm_epilogue_form<8, PERM, 0>(form, p, reinterpret_cast<f32x4(&)[4][8]>(acc[4]), m0 + 128 * wm, n0 + 128 * wn + 64, r16, quad);
            c_tile += tstep;
            fence();
        }
        hv_static_for<14>([&](auto B) __attribute__((always_inline)) {
            constexpr int b = decltype(B)::value;
            if constexpr (b < 8) wfB[b] = rd_w(ws, 1, b);
            xr[(b + 2) % 4] = rd_x(xs, (b + 2) / 8, (b + 2) % 8);
            issue_piece(HvInt<b + 2>{});
            mfma_block(B, HvInt<FIRST>{});
            fence();
        });
    };
    // (the first vmcnt(8) is trivially satisfied: the prologue's requests have landed)
    for (int t = 0; t < my_tiles; ++t) {
        round_step(HvInt<1>{}, t > 0, t > 0);
        for (int k = 1; k < nk; ++k) round_step(HvInt<0>{}, true, false);
    }
    // the last k-tile's tail blocks and the last tile's epilogue
    mfma_block(HvInt<14>{}, HvInt<0>{});
    mfma_block(HvInt<15>{}, HvInt<0>{});
    {
        int m0, n0;
        tile_origin(c_tile, m0, n0);
        hv_gemm_epilogue_form<8, PERM, 0>(form, p, reinterpret_cast<f32x4(&)[4][8]>(acc[0]), m0 + 128 * wm, n0 + 128 * wn, r16, quad);
        hv_gemm_epilogue_form<8, PERM, 0>(form, p, reinterpret_cast<f32x4(&)[4][8]>(acc[4]), m0 + 128 * wm, n0 + 128 * wn + 64, r16, quad);
    }
    hv_vm_wait<0>();  // (the streams' surplus pieces)
}
