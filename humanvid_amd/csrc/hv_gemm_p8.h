// hv_gemm_p8.h -- round 5: the 256 x 256 x 64 GEMM tile on deeper LDS rings (included by hv_gemm.h, which provides the
// epilogues, swizzles and the tile raster).  Reference call sites: the feed-forward input projections and the fused QKV
// projections of src/models/attention.py:298-443 / motion_module.py:157-175 (diffusers FeedForward / Attention).
//
// What bounded hv_gemm_glds_kernel<256,8,256,1> (profiles/r03_gemm_trace.txt, r04_s6): all eight waves run in lockstep, so the
// 24 fragment reads per wave and k-tile (192 KB per CU: 750 LDS cycles) and the eight LDS-DMA issues (60-180 cycles each) are
// phases in which no wave of the CU has an MFMA to issue, the k-tile behind the one being multiplied is the only one in flight
// (2-slot ring), and a k-step takes ~4 450 cycles for 2 086 cycles of MFMA issue.  Common to both schedules below:
//   * the whole 160 KiB of LDS: a 3-slot ring for the X tiles (96 KiB) and a 2-slot ring for the W tiles (64 KiB);
//   * per wave and k-tile: R1 (W kk0 + X rows [0,64) kk0: 8 reads) M1 | R2 (kk1: 8 reads) M2 | R3 (X rows [64,128) kk0: 4) M3 |
//     R4 (kk1: 4) M4 -- the MFMA order per accumulator is the old kernel's (kk0 then kk1): results are bit-identical;
//   * the LDS-DMA is issued from inline asm (hv_glds16_s): hipcc neither drains the queue in front of the fragment reads nor
//     turns their counted lgkmcnt waits into lgkmcnt(0);
//   * addressing: ONE per-lane byte offset per operand (row 8 wave + lane / 8 of a 64-row piece, swizzled chunk) that never
//     changes; everything that moves -- piece, k-tile, tile origin -- is added to the scalar base of the DMA instruction
//     (M % 64 == 0 and N % 64 == 0: a 64-row piece is either inside the matrix or replaced as a whole by the last valid one).
//
// SCHED = 0, "8 intervals" (cdna_hip_programming.md T3+T4+T5): the waves form two groups (waves 0-3 / 4-7: one wave of each on
// every SIMD) that run ONE barrier interval apart; a k-tile is eight intervals, in each of them one group multiplies (16 MFMAs
// per wave, wave priority raised) while the other reads its next fragments and issues its LDS-DMA.  The last read of a
// k-tile's W is in R2, of its X in R4, so W pieces can be issued one k-tile ahead and X pieces TWO: R1 issues W2 W3 of k-tile
// t+1, R2 X0 X2 of t+2, R3 X1 X3 of t+2, R4 W0 W1 of t+2, and ONE counted wait per k-tile (vmcnt(6)) in front of the barrier
// that opens the next k-tile covers both operands.  Tile ends: group 0 runs its epilogue in the interval in which group 1
// finishes the tile's last MFMAs and then runs its own -- the two epilogues overlap.
//   WAR / RAW, by global interval I(t, j) (k-tile t, j = 0..7; group 0 runs R1 M1 R2 M2 R3 M3 R4 M4 in j = 0..7, group 1 the
//   same one interval later): W slot t % 2 is last read in group 1's R2(t) = I(t,3), whose reads have returned before its M2
//   issues, i.e. before the barrier that ends I(t,4); the first write into it is W0 W1(t+2) from group 0's R4(t) = I(t,6).
//   X slot t % 3 is last read in group 1's R4(t) = I(t,7) (returned before the barrier that ends I(t+1,0)); the first write
//   into it is X0 X2(t+3) from group 0's R2(t+1) = I(t+1,2).  Every wave waits for its own pieces of k-tile t+1 before the
//   barrier that ends I(t,7) (group 0 after M4, group 1 after R4); group 0 first reads k-tile t+1 in I(t+1,0).
//
// SCHED = 1, "one barrier": the timing-only ablations of SCHED 0 (profiles/r05_s2.txt) say that its eight barriers per k-tile and
// its read / DMA-issue segments, not the data, are what it waits for.  With the same rings a single barrier per k-tile is
// enough for correctness when every piece of k-tile t+1 (W) and t+2 (X) is issued inside k-tile t and waited for in front of
// the barrier that ends it: W slot (t+1) % 2 and X slot (t+2) % 3 were last read in k-tile t-1, i.e. before the barrier that
// opened k-tile t.  The fragment reads of segment j+1 are issued in front of the MFMAs of segment j (the kk0 / kk1 register
// sets alternate: no extra registers); HV_P8_PRIO = 2 gives waves 4-7 a static priority so that the two waves of a SIMD fall out
// of step by themselves (one multiplies while the other issues).
#pragma once

#ifndef HV_P8_PRIO
#define HV_P8_PRIO 1
#endif
#ifndef HV_P8_ABL
#define HV_P8_ABL 0  // timing-only ablations (tools/build_variant.sh): 1 no LDS-DMA in the loop, 2 no fragment reads, 4 no MFMAs, 8 every piece from one hot 32 KiB, 32 no vmcnt waits
#endif
HV_DEV void hv_phase_barrier() {
#ifndef HV_EMU
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#else
    __syncthreads();
#endif
}
HV_DEV void hv_mfma_prio(int on) {
#ifndef HV_EMU
    if (on) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
#else
    (void)on;
#endif
}
#ifndef HV_EMU
#define P8_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define P8_FENCE() ((void)0)
#endif
#if !defined(HV_EMU) && (HV_P8_ABL & 2)
HV_DEV bf16x8 hv_p8_fake_read(const void* p) {  // timing build: no LDS read, an opaque register value
    unsigned u = (unsigned)(unsigned long)p;
    asm volatile("" : "+v"(u));
    return hv_as_bf16x8(u32x4{u, u, u, u});
}
#define P8_LD(ptr) hv_p8_fake_read(ptr)
#else
#define P8_LD(ptr) hv_as_bf16x8(hv_ld16(ptr))
#endif
#if !defined(HV_EMU) && (HV_P8_ABL & 4)
HV_DEV f32x4 hv_p8_fake_mfma(bf16x8 a, bf16x8 b, f32x4 c, int, int, int) {  // timing build: operands stay live, no MFMA
    asm volatile("" : "+v"(c) : "v"(a), "v"(b));
    return c;
}
#define P8_MFMA(...) hv_p8_fake_mfma(__VA_ARGS__)
#else
#define P8_MFMA(...) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__VA_ARGS__)
#endif

// LDS-DMA of one 1 KiB piece: scalar base (64-bit) + per-lane 32-bit byte offset -> wave-uniform LDS address.  The base is
// SALU arithmetic on kernel arguments almost everywhere; where hipcc has moved a (uniform) part of the chain to the VALU
// (the integer divisions of the tile raster), readfirstlane brings it back -- hence the s_nop 4 (an SGPR fresh from
// v_readfirstlane read as a VMEM base), which costs five cycles per piece.
HV_DEV void hv_glds16_p8(const void* base_uniform, unsigned byte_ofs, unsigned lds_addr_uniform) {
#ifndef HV_EMU
    const unsigned long a = (unsigned long)base_uniform;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned long b = ((unsigned long)hi << 32) | lo;
    const unsigned l = __builtin_amdgcn_readfirstlane(lds_addr_uniform);
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(byte_ofs), "s"(b), "s"(l)
                 : "memory");
#else
    (void)base_uniform, (void)byte_ofs, (void)lds_addr_uniform;
#endif
}

// GEGLU: the W-tile rows of a lane's fragments follow hv_perm_row_geglu (compile-time: the fragment offsets are immediates)
template <bool PERM, bool GEGLU, int SCHED>
__global__ __launch_bounds__(512, 2) void hv_gemm_p8_kernel(HvGemmParams p, int gm, int form) {
    constexpr int BM = 256, BN = 256, BK = 64, NW = 8;
    constexpr int WTM = 128, NMF = 8, HMF = 4;
    constexpr int XT = BM * BK * 2, WT = BN * BK * 2;  // 32 KiB each
    constexpr int XS = 3, WS = 2;                       // ring depths
    static_assert(!GEGLU || PERM, "GEGLU runs on the permuted channel assignment");
    __shared__ __attribute__((aligned(16))) unsigned char smem[XS * XT + WS * WT];  // 160 KiB: one workgroup per CU

    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifndef HV_EMU
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)smem;
#else
    const int wave = tid >> 6;
#endif
    const int wm = wave & 1, wn = wave >> 1;
    const int grp = wave >> 2;  // waves w and w + 4 share a SIMD (a workgroup's waves go round the four SIMDs)
    const int r16 = lane & 15, quad = lane >> 4;

    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int total = tiles_n * tiles_m;
    auto tile_origin = [&](int ti, int& m0, int& n0) __attribute__((always_inline)) {  // (raster: see hv_gemm_glds_kernel)
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
    const int nk = p.K / BK;
    const int nsteps = my_tiles * nk;

    // ---- issue state: two cursors over the flattened (tile, k-tile) sequence ----
    // piece q of a tile = rows [64 q, +64); DMA instruction (wave, q) covers rows 64 q + 8 wave + lane / 8.  Both swizzles only
    // look at row bits 0..3, so one per-lane offset per operand serves every piece.
    const int k1_steps = p.X2 != nullptr ? p.K1 / BK : -1;
    const int sub = lane >> 3, prow = 8 * wave + sub;  // row of the lane inside a 64-row piece
    const unsigned x_chunk = (unsigned)(((lane & 7) ^ ((prow >> 1) & 7)) * 16);
    const unsigned w_lane = (unsigned)prow * (unsigned)p.K * 2u +
                            (unsigned)(((lane & 7) ^ (PERM ? hv_wperm_swizzle(prow) : ((prow >> 1) & 7))) * 16);
    unsigned x_lane = 0;
    bool abl_prologue = true;  // (HV_P8_ABL timing builds only)
    int xi_tile = first, xi_k = 0, xi_slot = 0, xi_left = nsteps, xi_valid = 4;
    int wi_tile = first, wi_k = 0, wi_slot = 0, wi_left = nsteps, wi_valid = 4;
    // 32-bit scalar byte offsets from the operand bases (every operand spans < 4 GiB: hv_gemm_choose) -- gfx950 has no
    // 64-bit scalar multiply, and a 64-bit product would drag the whole address chain onto the VALU
    const char* x_src = reinterpret_cast<const char*>(p.X);
    unsigned x_koff = 0, w_koff = 0;  // piece 0 of the cursor's k-tile
    unsigned x_stride = 0;            // bytes between pieces
    const unsigned w_stride = 64u * (unsigned)p.K * 2u;
    int xi_m0 = 0;
    auto set_x_src = [&](bool second) __attribute__((always_inline)) {
        const unsigned ld = (unsigned)(second ? p.ldx2 : p.ldx);
        x_lane = (unsigned)prow * ld * 2u + x_chunk;
        x_stride = 64u * ld * 2u;
        x_src = reinterpret_cast<const char*>(second ? p.X2 : p.X);
        x_koff = (unsigned)xi_m0 * ld * 2u;
    };
    auto set_x_tile = [&]() __attribute__((always_inline)) {
        int n0;
        tile_origin(xi_tile, xi_m0, n0);
        xi_valid = min(4, (p.M - xi_m0) / 64);
        set_x_src(k1_steps == 0);
    };
    auto set_w_tile = [&]() __attribute__((always_inline)) {
        int m0, n0;
        tile_origin(wi_tile, m0, n0);
        wi_valid = min(4, (p.N - n0) / 64);
        w_koff = (unsigned)n0 * (unsigned)p.K * 2u;
    };
    auto issue_x = [&](auto Q) __attribute__((always_inline)) {
        constexpr int q = decltype(Q)::value;
        const int qq = q < xi_valid ? q : xi_valid - 1;  // a piece beyond the ragged M edge: the last valid one (results masked)
        if ((HV_P8_ABL & 1) && !abl_prologue) return;
#ifndef HV_EMU
        hv_glds16_p8(x_src + (x_koff + (unsigned)qq * x_stride), (HV_P8_ABL & 8) ? (x_lane & 0x7fffu) : x_lane, lds0 + xi_slot * XT + (wave + NW * q) * 1024);
#else
        memcpy(smem + xi_slot * XT + (wave + NW * q) * 1024 + lane * 16, x_src + (x_koff + (unsigned)qq * x_stride) + x_lane, 16);
#endif
    };
    auto issue_w = [&](auto Q) __attribute__((always_inline)) {
        constexpr int q = decltype(Q)::value;
        const int qq = q < wi_valid ? q : wi_valid - 1;
        if ((HV_P8_ABL & 1) && !abl_prologue) return;
#ifndef HV_EMU
        hv_glds16_p8(reinterpret_cast<const char*>(p.W) + (w_koff + (unsigned)qq * w_stride), (HV_P8_ABL & 8) ? (w_lane & 0x7fffu) : w_lane,
                     lds0 + XS * XT + wi_slot * WT + (wave + NW * q) * 1024);
#else
        memcpy(smem + XS * XT + wi_slot * WT + (wave + NW * q) * 1024 + lane * 16, reinterpret_cast<const char*>(p.W) + (w_koff + (unsigned)qq * w_stride) + w_lane, 16);
#endif
    };
    auto advance_x = [&]() __attribute__((always_inline)) {
        if (++xi_slot == XS) xi_slot = 0;
        --xi_left;
        x_koff += BK * 2;
        if (++xi_k == nk) {
            xi_k = 0;
            xi_tile += tstep;
            if (xi_left > 0) set_x_tile();
        } else if (xi_k == k1_steps) {
            set_x_src(true);
        }
    };
    auto advance_w = [&]() __attribute__((always_inline)) {
        if (++wi_slot == WS) wi_slot = 0;
        --wi_left;
        w_koff += BK * 2;
        if (++wi_k == nk) {
            wi_k = 0;
            wi_tile += tstep;
            if (wi_left > 0) set_w_tile();
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

    // prologue: k-tile 0 completely, then what the steady state would have issued before k-tile 0
    set_x_tile();
    set_w_tile();
    issue_w(HvInt<0>{});
    issue_w(HvInt<1>{});
    issue_w(HvInt<2>{});
    issue_w(HvInt<3>{});
    advance_w();
    issue_x(HvInt<0>{});
    issue_x(HvInt<2>{});
    issue_x(HvInt<1>{});
    issue_x(HvInt<3>{});
    advance_x();
    if (nsteps > 1) {
        issue_x(HvInt<0>{});
        issue_x(HvInt<2>{});
        issue_x(HvInt<1>{});
        issue_x(HvInt<3>{});
        advance_x();
        if (SCHED == 0) {
            issue_w(HvInt<0>{});
            issue_w(HvInt<1>{});
            hv_vm_wait<6>();
        } else {
            hv_vm_wait<4>();
        }
    } else {
        hv_vm_wait<0>();
    }
    hv_phase_barrier();
    if (SCHED == 0 && grp == 1) hv_phase_barrier();  // group 1 runs one interval behind
    abl_prologue = false;
#ifndef HV_EMU
    if (SCHED == 1 && HV_P8_PRIO == 2 && grp == 1) __builtin_amdgcn_s_setprio(1);
#endif

    // lane constants of the fragment reads (byte offsets inside a tile); the other fragments are compile-time offsets away:
    // X rows + 16 f: + 2048 f bytes (the swizzle sees row / 2 % 8); W rows + dW(f) with dW a multiple of 4 that leaves the row
    // bits 0, 1, 3 of hv_wperm_swizzle (and row / 2 % 8 of the plain swizzle: dW = 16 f) alone
    int xoff[2], woff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        xoff[kk] = hv_swz<BK>(WTM * wm + r16, kk * 4 + quad);
        const int wrow0 = !PERM ? r16 : (GEGLU ? hv_perm_row_geglu(0, r16) : hv_perm_row(0, r16));
        woff[kk] = PERM ? hv_swz_wperm(64 * wn + wrow0, kk * 4 + quad) : hv_swz<BK>(64 * wn + wrow0, kk * 4 + quad);
    }
    auto wdelta = [](int f) constexpr { return 128 * (!PERM ? 16 * f : (GEGLU ? 16 * (f & 1) + 4 * (f >> 1) : 32 * (f >> 1) + 4 * (f & 1))); };

    bf16x8 wf[2][4], xa[HMF], xb[HMF];
    const unsigned char* xs = smem;
    const unsigned char* ws = smem + XS * XT;
    int c_tile = first, c_k = 0, cx_slot = 0, cw_slot = 0;
    auto next_slots = [&]() __attribute__((always_inline)) {
        if (++cx_slot == XS) cx_slot = 0;
        if (++cw_slot == WS) cw_slot = 0;
        xs = smem + cx_slot * XT;
        ws = smem + XS * XT + cw_slot * WT;
    };
    auto read_w = [&](int kk) __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < 4; ++f) wf[kk][f] = P8_LD(ws + woff[kk] + wdelta(f));
    };
    auto read_x = [&](bf16x8(&x)[HMF], int kk, int half) __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < HMF; ++f) x[f] = P8_LD(xs + xoff[kk] + 2048 * (HMF * half + f));
    };
    auto mfma16 = [&](bf16x8(&w)[4], bf16x8(&x)[HMF], int mbase) __attribute__((always_inline)) {
        if (SCHED == 1) P8_FENCE();  // (the reads / DMA issued in front stay in front)
        if (HV_P8_PRIO == 1) hv_mfma_prio(1);
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
            for (int mf = 0; mf < HMF; ++mf) acc[nf][mbase + mf] = P8_MFMA(w[nf], x[mf], acc[nf][mbase + mf], 0, 0, 0);
        if (HV_P8_PRIO == 1) hv_mfma_prio(0);
        if (SCHED == 1) P8_FENCE();
    };
    auto wait_pieces = [&](bool has2, auto N) __attribute__((always_inline)) {
        if (HV_P8_ABL & 32) return;
        if (has2) hv_vm_wait<decltype(N)::value>();
        else hv_vm_wait<0>();
    };
    auto epilogue = [&]() __attribute__((always_inline)) {
        int m0, n0;
        tile_origin(c_tile, m0, n0);
#ifdef HV_GEMM_TRACE
        int hv_ti = 8192;
#endif
        hv_gemm_epilogue_form<NMF, PERM, 0>(form, p, acc, m0 + WTM * wm, n0 + 64 * wn, r16, quad HV_TRACE_ARG);
        c_tile += tstep;
        clear_acc();
    };

    if constexpr (SCHED == 1) {
        read_w(0);
        read_x(xa, 0, 0);
        for (int t = 0; t < nsteps; ++t) {
            const bool has1 = t + 1 < nsteps, has2 = t + 2 < nsteps;
            // R2 reads + W0 W1 of k-tile t+1, in front of M1
            read_w(1);
            read_x(xb, 1, 0);
            if (has1) {
                issue_w(HvInt<0>{});
                issue_w(HvInt<1>{});
            }
            mfma16(wf[0], xa, 0);
            // R3 reads + W2 W3 of k-tile t+1, in front of M2
            read_x(xa, 0, 1);
            if (has1) {
                issue_w(HvInt<2>{});
                issue_w(HvInt<3>{});
                advance_w();
            }
            mfma16(wf[1], xb, 0);
            // R4 reads + X0 X2 of k-tile t+2, in front of M3
            read_x(xb, 1, 1);
            if (has2) {
                issue_x(HvInt<0>{});
                issue_x(HvInt<2>{});
            }
            mfma16(wf[0], xa, HMF);
            // X1 X3 of k-tile t+2; this wave's pieces of k-tile t+1 have landed; every read of k-tile t has returned
            if (has2) {
                issue_x(HvInt<1>{});
                issue_x(HvInt<3>{});
                advance_x();
            }
            wait_pieces(has2, HvInt<4>{});
            hv_barrier_raw();  // (lgkmcnt(0) + s_barrier)
            next_slots();
            const bool tile_end = ++c_k == nk;
            if (!tile_end && has1) {  // R1 of k-tile t+1 in front of M4 (at a tile end: behind the epilogue)
                read_w(0);
                read_x(xa, 0, 0);
            }
            mfma16(wf[1], xb, HMF);
            if (tile_end) {
                c_k = 0;
                epilogue();
                if (has1) {
                    read_w(0);
                    read_x(xa, 0, 0);
                }
            }
        }
        return;
    }

    for (int t = 0; t < nsteps; ++t) {
        const bool has1 = t + 1 < nsteps, has2 = t + 2 < nsteps;
        // ---- R1: W kk0, X rows [0, 64) kk0; issue W2 W3 of k-tile t+1
        read_w(0);
        read_x(xa, 0, 0);
        if (has1) {
            issue_w(HvInt<2>{});
            issue_w(HvInt<3>{});
            advance_w();
        }
        hv_phase_barrier();
        mfma16(wf[0], xa, 0);  // ---- M1
        hv_phase_barrier();
        // ---- R2: W kk1, X rows [0, 64) kk1; issue X0 X2 of k-tile t+2
        read_w(1);
        read_x(xb, 1, 0);
        if (has2) {
            issue_x(HvInt<0>{});
            issue_x(HvInt<2>{});
        }
        hv_phase_barrier();
        mfma16(wf[1], xb, 0);  // ---- M2
        hv_phase_barrier();
        // ---- R3: X rows [64, 128) kk0; issue X1 X3 of k-tile t+2
        read_x(xa, 0, 1);
        if (has2) {
            issue_x(HvInt<1>{});
            issue_x(HvInt<3>{});
            advance_x();
        }
        hv_phase_barrier();
        mfma16(wf[0], xa, HMF);  // ---- M3
        hv_phase_barrier();
        // ---- R4: X rows [64, 128) kk1; issue W0 W1 of k-tile t+2; group 1 waits for its pieces of k-tile t+1 here
        read_x(xb, 1, 1);
        if (has2) {
            issue_w(HvInt<0>{});
            issue_w(HvInt<1>{});
        }
        if (grp == 1) wait_pieces(has2, HvInt<6>{});
        hv_phase_barrier();
        mfma16(wf[1], xb, HMF);  // ---- M4
        if (grp == 0) {  // ... group 0 here, in front of the barrier that opens k-tile t+1 for it
            wait_pieces(has2, HvInt<6>{});
            hv_phase_barrier();
        }
        next_slots();
        if (++c_k == nk) {
            // tile end: group 0 is behind its barrier, group 1 in front of it -- both epilogues run in the same interval
            c_k = 0;
            epilogue();
        }
        if (grp == 1) hv_phase_barrier();
    }
    if (grp == 0) hv_phase_barrier();  // pairs with group 1's last barrier
}
