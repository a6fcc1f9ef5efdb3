// hv_conv.h -- 3x3 convolution as an implicit GEMM on MFMA with an LDS-staged (y,x) halo tile.
//
// Reference ops replaced: InflatedConv3d 3x3 (/root/reference/src/models/resnet.py:9-15), with
// the GroupNorm-apply + SiLU that precedes it in ResnetBlock3D (resnet.py:218-221, 232-238) fused
// into the tile load, the time-embedding add (resnet.py:224-229) / residual add (resnet.py:243) /
// pose-feature add (src/models/unet_3d.py:482-484) fused into the epilogue, Downsample3D stride 2
// (resnet.py:110-118), Upsample3D nearest-2x (resnet.py:51-88) folded into the tile addressing and
// the skip-connection torch.cat (src/models/unet_3d_blocks.py:698,828) as a two-source channel loop.
//
// Design (MI355X): one workgroup = 128 output pixels (TH x TW patch of one image) x 128 output
// channels; 4 waves (2 pixel halves x 2 channel halves), 4x4 MFMA 16x16x32 fragments per wave.
// The reduction runs over (32-channel chunk, tap): per chunk the (TH+2)x(TW+2) input halo is
// loaded ONCE (coalesced 16-byte reads along channels), normalised/activated once and parked in
// LDS; the nine taps then read shifted windows of it, so the activation is fetched ~1.4x instead
// of 9x and GroupNorm/SiLU cost 1/9 of a per-tap scheme.  Weight tiles [128][32] stream per tap,
// double-buffered through registers so one barrier separates K-steps.  LDS pixel stride is 80 B
// (64 B data + 16 B pad) which makes the 16 lanes of a ds_read_b128 group conflict-free.
#pragma once
#include "hv_common.h"
#include "hv_gemm.h"  // hv_swz
#include "humanvid_hip.h"

#ifndef HV_CONV_HBUFS1
#define HV_CONV_HBUFS1 1  // one halo buffer for every variant (0: two for CK = 32, the round-2 layout; same-box A/B in profiles/r03_conv_lds_ab.txt)
#endif
#ifndef HV_CONV_PITCH32
#define HV_CONV_PITCH32 1  // 0: the round-2 LDS pitches / weight swizzle (A/B builds)
#endif
// Weight-tile swizzle.  CK = 64 (128-byte rows): the GEMM's hv_swz<64>.  CK = 32 (64-byte rows, four rows per 256-byte bank
// row): chunk ^= f((row >> 2) & 3) with f = {0, 2, 3, 1} -- hv_swz<32> (f = identity) puts the two quads that a ds_read_b128
// lane group mixes on the same banks (2-way conflict on every weight-fragment read); this f is conflict-free for all four
// groups and all four fragments.
template <int CK>
HV_DEV int hv_swz_conv_f(int row) {
    if constexpr (CK == 64 || !HV_CONV_PITCH32) return (row / (16 / (CK / 8))) % (CK / 8);
    else {
        const int b = (row >> 2) & 3;
        return ((b & 1) << 1) ^ ((b >> 1) * 3);
    }
}
template <int CK>
HV_DEV int hv_swz_conv(int row, int chunk) {
    return row * (CK * 2) + ((chunk ^ hv_swz_conv_f<CK>(row)) << 4);
}

template <int TW, int MODE, int NPIX = 128, int WPX = 64, int CK = 32>
struct HvConvGeom {
    static constexpr int NT = 2 * 64 * (NPIX / WPX);   // threads: one wave per WPX pixels x 64 channels
    static constexpr int TH = NPIX / TW;
    static constexpr int HH = MODE == HV_CONV_S1 ? TH + 2 : (MODE == HV_CONV_S2 ? 2 * TH + 1 : TH / 2 + 2);
    static constexpr int HW = MODE == HV_CONV_S1 ? TW + 2 : (MODE == HV_CONV_S2 ? 2 * TW + 1 : TW / 2 + 2);
    static constexpr int HP = HH * HW;
    static constexpr int CPP = CK / 8;        // 16-byte pieces per pixel and channel chunk
    // bytes per halo pixel in LDS: data + pad.  A ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19,
    // 28-31}, ... (MI355X_MICROARCH.md): a group mixes two quads of a fragment read -- pixels r16 at byte 16 q and pixels r16'
    // at 16 (q + 1).  The "odd multiple of 16 bytes" pitch of rounds 1-2 (80 / 144) is a 2-way bank conflict on EVERY fragment
    // read under that grouping (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50 measured, profiles/r03_lds_conflicts.txt:
    // half of the LDS cycles of a kernel that issues one fragment read per two MFMAs).  Conflict-free pitches, checked per
    // lane group with the documented bank rule (the model that reproduces the GEMM's zero): = 32 (mod 64) bytes for the
    // consecutive pixels of a stride-1 tap window (96 / 160), the old pitch for the every-other-pixel reads of the stride-2
    // form (2 x 80 = 160), 112 for the pixel pairs of the upsample-folded form.
    static constexpr int PS = HV_CONV_PITCH32 ? CK * 2 + (MODE == HV_CONV_S1 ? 32 : (MODE == HV_CONV_S2 ? 16 : 48)) : CK * 2 + 16;
    static constexpr int HALO_BYTES = ((HP * PS + 15) / 16) * 16;
    static constexpr int HALO_ITERS = (HP * CPP + NT - 1) / NT;
    static constexpr int WTILE_BYTES = 128 * CK * 2;
    static constexpr int HBUFS = (CK == 64 || HV_CONV_HBUFS1) ? 1 : 2;  // one halo buffer (an extra barrier per chunk) keeps the workgroups per CU
};

// GLDS = true: the weight tiles stream HBM/L2 -> LDS with global_load_lds into a 3-slot ring (two
// taps in flight across one raw barrier per step, counted vmcnt); only the halo tile, which needs the
// GroupNorm/SiLU transform, is still staged through registers (once per 9 steps).
// NPIX = 256 (8 waves): the weight tile of a tap is shared by twice as many pixels -> ~40 % fewer
// bytes per FLOP through the per-CU load path, which bounds the 128-pixel variant (~25 GB/s per CU).
// WPX = 128 (round 2, NPIX = 256 on 4 waves): a wave owns 128 pixels x 64 channels (8 x 4 MFMA fragments, 128 accumulator
// registers) -- 32 MFMAs per 12 ds_read_b128 and per barrier instead of 16 per 8, the weight tile of a tap serves twice the
// pixels, the 18 x 18 halo of a 16 x 16 patch carries 27 % border instead of 41 %; 76 KiB of LDS, two workgroups per CU.
// CK = 64 (round 2, opt-in: HV_TUNE_CONV_BIG = 3; stride 1, C1 and C2 multiples of 64): the reduction runs over 64-channel
// chunks -- the weight rows of a tap are whole 128-byte lines (the LDS-DMA path moves 64-byte row segments at half the rate:
// the finding behind the GEMM's BK = 64), every tap step carries 32 MFMAs per wave behind its barrier instead of 16, and the
// number of steps, barriers and DMA instructions halves.  74 KiB of LDS (one 26 KiB halo buffer, three 16 KiB weight slots).
template <int TW, int MODE, bool GLDS, int NPIX, int WPX = 64, int CK = 32>
__global__ __launch_bounds__(2 * 64 * (NPIX / WPX), (WPX == 128 ? 2 : 1)) void hv_conv3x3_kernel(hv_conv3x3_params p, int raster) {
    using G = HvConvGeom<TW, MODE, NPIX, WPX, CK>;
    static_assert(CK == 32 || (CK == 64 && GLDS), "64-channel chunks stream their weights by LDS-DMA");
    constexpr int TH = G::TH;
    constexpr int NT = G::NT, NW = NT / 64, WM = NPIX / WPX, NMF = WPX / 16;
    constexpr int CPP = G::CPP, CPP_SH = CK == 64 ? 3 : 2;  // shifts / masks: lane and thread indices are non-negative
    constexpr int WSLOTS = GLDS ? 3 : 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::HBUFS * G::HALO_BYTES + WSLOTS * G::WTILE_BYTES];
    unsigned char* halo = smem;
    unsigned char* wsm = smem + G::HBUFS * G::HALO_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM, r16 = lane & 15, quad = lane >> 4;
    const int Cin = p.C1 + p.C2;

    const int tiles_n = (p.Cout + 127) / 128;
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    const int total = p.n_images * tiles_y * tiles_x * tiles_n;
    const int cpx = gridDim.x / 8;
    int t = (blockIdx.x % 8) * cpx + blockIdx.x / 8;
    if (t >= total) return;
    // Workgroup raster inside an XCD's contiguous range of t.  0: the output-channel tiles of a pixel patch are adjacent (its
    // halo is fetched once and re-read from L2 by the tiles_n workgroups; the XCD's concurrent workgroups then stream ALL
    // weight tiles, 29 MB at 1280 -> 1280 channels against 4 MB of L2: profiles/r03_pmc_traffic.json counts the weights
    // 15-24 x).  1: the pixel patches of ONE output-channel tile are adjacent (its 9 x Cin x 128 weights stay in L2 while the
    // XCD walks the patches; every patch's halo is fetched once per channel tile instead).
    int n0;
    if (raster == 0) {
        n0 = (t % tiles_n) * 128;
        t /= tiles_n;
    } else {
        const int npatch = p.n_images * tiles_y * tiles_x;
        n0 = (t / npatch) * 128;
        t %= npatch;
    }
    const int x0 = (t % tiles_x) * TW;
    t /= tiles_x;
    const int y0 = (t % tiles_y) * TH;
    const int img = t / tiles_y;

    int in_y0, in_x0;
    if (MODE == HV_CONV_S1) {
        in_y0 = y0 - 1;
        in_x0 = x0 - 1;
    } else if (MODE == HV_CONV_S2) {
        in_y0 = 2 * y0 - 1;
        in_x0 = 2 * x0 - 1;
    } else {
        in_y0 = y0 / 2 - 1;
        in_x0 = x0 / 2 - 1;
    }

    const int nchunks = Cin / CK;
    const int nsteps = nchunks * 9;
    const int hc = tid & (CPP - 1);  // this thread's 16-byte channel slot inside a chunk (constant: NT % CPP == 0)

    u32x4 hreg[G::HALO_ITERS];
    constexpr int WIT = (128 * CPP) / NT;  // 16-byte chunks of a weight tile per thread
    u32x4 wreg[WIT];

    auto load_halo = [&](int chunk) {
        const int ci = chunk * CK + hc * 8;
        const bool second = ci >= p.C1;
        const bf16_t* base = second ? p.X2 : p.X;
        const int cs = second ? p.C2 : p.C1;
        const int cc = second ? ci - p.C1 : ci;
#pragma unroll
        for (int j = 0; j < G::HALO_ITERS; ++j) {
            const int hp = (tid + NT * j) >> CPP_SH;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (hp < G::HP) {
                const int iy = in_y0 + hp / G::HW, ix = in_x0 + hp % G::HW;
                if (iy >= 0 && iy < p.Hs && ix >= 0 && ix < p.Ws)
                    v = hv_ld16(base + ((long)(img * p.Hs + iy) * p.Ws + ix) * cs + cc);
            }
            hreg[j] = v;
        }
    };

    auto store_halo = [&](int chunk, int buf) {
        const int ci = chunk * CK + hc * 8;
        float sc[8], sh[8];
        const bool pro = p.pro_scale != nullptr;
        if (pro) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sc[e] = p.pro_scale[(long)img * Cin + ci + e];
                sh[e] = p.pro_shift[(long)img * Cin + ci + e];
            }
        }
#pragma unroll
        for (int j = 0; j < G::HALO_ITERS; ++j) {
            const int hp = (tid + NT * j) >> CPP_SH;
            if (hp >= G::HP) continue;
            u32x4 v = hreg[j];
            if (pro || p.pro_act != HV_ACT_NONE) {
                const int iy = in_y0 + hp / G::HW, ix = in_x0 + hp % G::HW;
                if (iy >= 0 && iy < p.Hs && ix >= 0 && ix < p.Ws) {  // zero padding stays zero
                    float f[8];
                    hv_unpack8(v, f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (pro) f[e] = f[e] * sc[e] + sh[e];
                        f[e] = hv_act(f[e], p.pro_act);
                    }
                    v = hv_pack8(f);
                }
            }
            hv_st16(halo + buf * G::HALO_BYTES + hp * G::PS + hc * 16, v);
        }
    };

    auto load_w = [&](int step) {
        const int chunk = step / 9, tap = step % 9;
#pragma unroll
        for (int i = 0; i < WIT; ++i) {
            const int id = tid + NT * i;
            const int row = id >> CPP_SH, c = id & (CPP - 1);
            const int n = n0 + row;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (n < p.Cout) v = hv_ld16(p.W + ((long)n * 9 + tap) * Cin + chunk * CK + c * 8);
            wreg[i] = v;
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WIT; ++i) {
            const int id = tid + NT * i;
            hv_st16(wsm + buf * G::WTILE_BYTES + hv_swz_conv<CK>(id >> CPP_SH, id & (CPP - 1)), wreg[i]);
        }
    };
    // LDS-DMA form: 8 wave-instructions of 1 KiB (16 rows x 64 B) per tap tile, 8 / NW per wave; the swizzle of
    // hv_swz<32> is applied on the source address; rows beyond Cout are clamped.  The per-lane part of the source address
    // (weight row, swizzled chunk) is computed once (wofs, 32-bit byte offsets: hv_conv3x3_launch checks the span), the
    // per-step part (tap, channel chunk) is a scalar added to the base; the copies are issued from inline asm
    // (hv_glds16_s) so that hipcc's wait-count tracker does not put a vmcnt(0) in front of every ds_read of the step.
    constexpr int RPI = 1024 / (CK * 2);            // weight rows per 1 KiB wave-instruction: 16 (CK = 32) / 8 (CK = 64)
    constexpr int WQ = (G::WTILE_BYTES / 1024) / NW;  // DMA instructions per wave and tap tile
    static_assert(WQ >= 1 && WQ <= 4, "one to four LDS-DMA instructions per wave and tap");
#ifndef HV_EMU
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#else
    const int wave_u = wave;
#endif
    unsigned wofs0 = 0, wofs1 = 0, wofs2 = 0, wofs3 = 0;
    {
        const int sub = lane >> CPP_SH, pc = lane & (CPP - 1);
        constexpr int RPB_SH = CK == 64 ? 1 : 2;  // log2(rows per 256-byte bank row): the swizzle of hv_swz<CK> on the source side
        hv_static_for<WQ>([&](auto Q) __attribute__((always_inline)) {
            constexpr int q = decltype(Q)::value;
            const int row = RPI * (wave_u + NW * q) + sub;
            hv_pick4<q>(wofs0, wofs1, wofs2, wofs3) =
                ((unsigned)min(n0 + row, p.Cout - 1) * 9u * (unsigned)Cin + (unsigned)((pc ^ hv_swz_conv_f<CK>(row)) * 8)) * 2u;
        });
    }
    int iw_chunk = 0, iw_tap = 0, iw_slot = 0;  // issue state: advanced one step per call
    auto issue_w = [&]() __attribute__((always_inline)) {
        unsigned char* slot = wsm + iw_slot * G::WTILE_BYTES;
        const unsigned step = (unsigned)(iw_tap * Cin + iw_chunk * CK) * 2u;  // wave-uniform part of the offset
        hv_static_for<WQ>([&](auto Q) __attribute__((always_inline)) {
            constexpr int q = decltype(Q)::value;
            // base = the kernel argument: always an SGPR pair
            hv_glds16_s(p.W, hv_pick4<q>(wofs0, wofs1, wofs2, wofs3) + step, slot + (wave_u + NW * q) * 1024);
        });
        if (++iw_slot == 3) iw_slot = 0;
        if (++iw_tap == 9) {
            iw_tap = 0;
            ++iw_chunk;
        }
    };

    // per-lane pixel coordinates of the NMF pixel fragments
    int py[NMF], px[NMF];
#pragma unroll
    for (int mf = 0; mf < NMF; ++mf) {
        const int pix = WPX * wm + 16 * mf + r16;
        py[mf] = pix / TW;
        px[mf] = pix % TW;
    }
    // halo-pixel index of tap (0, 0) per fragment (stride-1 / stride-2 forms: the taps are immediate offsets from it)
    int lpb[NMF];
#pragma unroll
    for (int mf = 0; mf < NMF; ++mf)
        lpb[mf] = MODE == HV_CONV_S2 ? 2 * py[mf] * G::HW + 2 * px[mf] : py[mf] * G::HW + px[mf];

    f32x4 acc[4][NMF];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NMF; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_halo(0);
    if (GLDS) {
        issue_w();
        if (nsteps > 1) issue_w();
    } else {
        load_w(0);
    }
    // k-loop: channel chunks x the nine taps, the taps as a compile-time loop.  (As one runtime loop over (chunk, tap)
    // steps -- the round-1 form -- the halo registers loaded at tap 8 for the next chunk are live across the back-edge of
    // EVERY step; hipcc's wait-count tracker then protects them with vmcnt(3) / (1) / (0) in front of each step's ds_reads,
    // which drained the LDS-DMA weight ring in every step.  Unrolled, the only compiler waits left are the true ones at
    // tap 0, and the tap offsets / ring slots (9 % 3 == 0) are immediates.)
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int hbuf = G::HBUFS == 2 ? (chunk & 1) : 0;
        hv_static_for<9>([&](auto T) __attribute__((always_inline)) {
            constexpr int tap = decltype(T)::value;
            const int s = chunk * 9 + tap;
            const int wbuf = GLDS ? tap % 3 : (s & 1);
            if (tap == 0) {
                // one halo buffer: every wave must be done with the previous chunk's tap-8 reads before it is overwritten
                if (G::HBUFS == 1 && chunk > 0) hv_barrier_raw();
                store_halo(chunk, hbuf);
            }
            if (GLDS) {
                if (s + 1 < nsteps)
                    hv_vm_wait<WQ>();  // tap tile s landed, tile s+1 may stay in flight
                else
                    hv_vm_wait<0>();
                hv_barrier_raw();
                if (s + 2 < nsteps) issue_w();  // k-step s+2: reuses the slot read at step s-1
                if (tap == 8 && s + 1 < nsteps) load_halo(chunk + 1);
            } else {
                store_w(wbuf);
                __syncthreads();
                if (s + 1 < nsteps) {
                    load_w(s + 1);
                    if (tap == 8) load_halo(chunk + 1);
                }
            }
            constexpr int dy = tap / 3, dx = tap - dy * 3;
            const unsigned char* hb = halo + hbuf * G::HALO_BYTES + quad * 16;
            const unsigned char* wb = wsm + wbuf * G::WTILE_BYTES;
#pragma unroll
            for (int kk = 0; kk < CK / 32; ++kk) {  // 32-channel MFMA slices of the chunk
                bf16x8 wf[4];
#pragma unroll
                for (int f = 0; f < 4; ++f)
                    wf[f] = hv_as_bf16x8(hv_ld16(wb + hv_swz_conv<CK>(64 * wn + 16 * f + r16, kk * 4 + quad)));
                // pixel fragments in groups of four (16 registers of operands at a time)
#pragma unroll
                for (int g = 0; g < NMF; g += 4) {
                    bf16x8 xf[4];
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        int lp;
                        if (MODE == HV_CONV_S1)
                            lp = lpb[g + f] + dy * G::HW + dx;
                        else if (MODE == HV_CONV_S2)
                            lp = lpb[g + f] + dy * G::HW + dx;
                        else
                            lp = ((py[g + f] + dy + 1) >> 1) * G::HW + ((px[g + f] + dx + 1) >> 1);
                        xf[f] = hv_as_bf16x8(hv_ld16(hb + lp * G::PS + kk * 64));
                    }
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                        for (int mf = 0; mf < 4; ++mf)
                            acc[nf][g + mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nf], xf[mf], acc[nf][g + mf], 0, 0, 0);
                }
            }
        });
    }

    // ---- epilogue, one 16-channel fragment column (nf) at a time: its bias / time-embedding vector, the residual fragments
    // of all pixel rows (loads batched before their first use: see the note on hv_gemm_epilogue), arithmetic + stores, and --
    // optional, p.gn_part -- the GroupNorm partial statistics of what was stored: per channel the sum and the sum of squares
    // over this wave's valid pixels, in-lane over the pixel fragments, then over the 16 lanes of the DPP row.  (Column-major
    // order keeps eight statistics registers live instead of 32: the row-major form of this epilogue with the statistics
    // came out at 304 registers.)
    const float* rv = p.rowvec ? p.rowvec + (long)(img / p.images_per_rowvec) * p.rowvec_ld : nullptr;
    const int rimg = p.residual ? (p.residual_images > 0 ? img % p.residual_images : img) : 0;
    const int gn_parts = tiles_y * tiles_x * WM;
    float* const gn_dst = p.gn_part ? p.gn_part + ((long)img * gn_parts + ((y0 / TH) * tiles_x + x0 / TW) * WM + wm) * p.Cout * 2
                                    : nullptr;
    // The stores go out in a second, ROW-major pass (pixel fragment outer, the four 32-byte channel segments of a pixel's
    // 128-byte line back to back): issued column by column -- a line's four segments a whole column pass apart -- the
    // partial lines reached HBM separately (WRITE_SIZE 2.3x the algorithmic bytes at 320 -> 320 channels,
    // profiles/r03_pmc_traffic.json before this change).  The packed results replace the accumulators (32 registers).
    u32x2 outp[4][NMF];
    long opix[NMF];
#pragma unroll
    for (int mf = 0; mf < NMF; ++mf) {
        const int opx = WPX * wm + 16 * mf + r16;  // recomputed: py / px need not stay live through the k-loop
        const int oy = y0 + opx / TW, ox = x0 + opx % TW;
        opix[mf] = (oy < p.Ho && ox < p.Wo) ? (long)(img * p.Ho + oy) * p.Wo + ox : -1;
    }
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
        const int n = n0 + 64 * wn + 16 * nf + 4 * quad;
        if (n >= p.Cout) continue;
        f32x4 add = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) add += *reinterpret_cast<const f32x4*>(p.bias + n);
        if (rv) add += *reinterpret_cast<const f32x4*>(rv + n);
        u32x2 res2[NMF];
#pragma unroll
        for (int mf = 0; mf < NMF; ++mf) {
            u32x2 r = {0u, 0u};
            if (p.residual && opix[mf] >= 0) {
                const long rpix = p.residual_images > 0 ? opix[mf] - (long)(img - rimg) * p.Ho * p.Wo : opix[mf];
                r = hv_ld8(p.residual + rpix * p.Cout + n);
            }
            res2[mf] = r;
        }
        f32x4 gs = {0.f, 0.f, 0.f, 0.f}, gq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mf = 0; mf < NMF; ++mf) {
            if (opix[mf] < 0) continue;
            f32x4 v = acc[nf][mf] + add;
            v[0] += hv_bf2f((bf16_t)(res2[mf][0] & 0xffff));
            v[1] += hv_bf2f((bf16_t)(res2[mf][0] >> 16));
            v[2] += hv_bf2f((bf16_t)(res2[mf][1] & 0xffff));
            v[3] += hv_bf2f((bf16_t)(res2[mf][1] >> 16));
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = hv_act(v[r], p.out_act);
            outp[nf][mf] = u32x2{hv_pack2(v[0], v[1]), hv_pack2(v[2], v[3])};
            gs += v;
            gq += v * v;
        }
        if (gn_dst != nullptr) {  // (wave-uniform)
            f32x4 a, b;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] = hv_row16_sum(gs[e]);
                b[e] = hv_row16_sum(gq[e]);
            }
            if (r16 == 0) {  // channels n .. n+3: {sum, sumsq} interleaved
                *reinterpret_cast<f32x4*>(gn_dst + 2 * n) = f32x4{a[0], b[0], a[1], b[1]};
                *reinterpret_cast<f32x4*>(gn_dst + 2 * n + 4) = f32x4{a[2], b[2], a[3], b[3]};
            }
        }
    }
#pragma unroll
    for (int mf = 0; mf < NMF; ++mf) {
        if (opix[mf] < 0) continue;
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const int n = n0 + 64 * wn + 16 * nf + 4 * quad;
            if (n >= p.Cout) continue;
            hv_st8(p.Y + opix[mf] * p.Cout + n, outp[nf][mf]);
        }
    }
}

#include "hv_conv4.h"  // hv_conv_w4_kernel: 12 x 16 pixels x 320 channels on four waves (plain single-source stride-1 inputs)

static int g_hv_conv_glds = 1;  // tuning knob (hv_set_tuning): LDS-DMA weight tiles

static int g_hv_conv_big = 1;  // tuning knob: 256-pixel tiles (8 waves) on images that fill them
static int g_hv_conv_raster = 2;  // tuning knob (hv_set_tuning key 9): 0 / 1 always that raster, 2 = raster 1 where the weights exceed the XCD's L2 (Cin x Cout >= 640 x 640)

template <int TW, int MODE, int NPIX, int WPX = 64, int CK = 32>
static inline void hv_conv3x3_launch_t(const hv_conv3x3_params& p, hipStream_t stream) {
    constexpr int TH = NPIX / TW;
    constexpr int NT = 2 * 64 * (NPIX / WPX);
    const int tiles = p.n_images * ((p.Ho + TH - 1) / TH) * ((p.Wo + TW - 1) / TW) * ((p.Cout + 127) / 128);
    const int grid = ((tiles + 7) / 8) * 8;
    hv_note("hv_conv3x3_kernel<%d,%d,%d,%d%s%s> | n=%d Hs=%d Ws=%d Ho=%d Wo=%d Cin=%d Cout=%d gn=%d res=%d", TW, MODE,
            g_hv_conv_glds, NPIX, WPX == 128 ? ",128" : "", CK == 64 ? ",ck64" : "", p.n_images, p.Hs, p.Ws, p.Ho, p.Wo, p.C1 + p.C2,
            p.Cout, p.pro_scale != nullptr, p.residual != nullptr);
    const int raster = g_hv_conv_raster == 2 ? ((long)(p.C1 + p.C2) * p.Cout >= 640L * 640 ? 1 : 0) : g_hv_conv_raster;
    if (g_hv_conv_glds || WPX == 128 || CK == 64)
        hv_launch(hv_conv3x3_kernel<TW, MODE, true, NPIX, WPX, CK>, dim3(grid), dim3(NT), stream, p, raster);
    else if constexpr (CK == 32)
        hv_launch(hv_conv3x3_kernel<TW, MODE, false, NPIX, WPX, CK>, dim3(grid), dim3(NT), stream, p, raster);
}

// (tile rows, tile columns, pixel halves per tile) of the kernel hv_conv3x3_launch selects for this problem: the layout of
// gn_part is [n_images][tiles_y * tiles_x * WM][Cout][2]
static inline void hv_conv3x3_tile_shape(const hv_conv3x3_params& p, int& TH, int& TW, int& WM) {
    if (hv_conv_w4_applies(p)) {
        TH = HvConv4Geom<>::TH, TW = HvConv4Geom<>::TW, WM = HvConv4Geom<>::WM;
        return;
    }
    const bool narrow = p.Wo <= 8;
    const bool big = g_hv_conv_big && !narrow && p.Ho >= 16 && p.mode == HV_CONV_UP2;
    TW = narrow ? 8 : 16;
    const int npix = big ? 256 : 128;
    TH = npix / TW;
    WM = npix / 64;
}
static inline int hv_conv3x3_gn_parts_of(const hv_conv3x3_params& p) {
    if (p.Cout % 4 != 0 || p.Ho <= 0 || p.Wo <= 0) return 0;
    int TH, TW, WM;
    hv_conv3x3_tile_shape(p, TH, TW, WM);
    return ((p.Ho + TH - 1) / TH) * ((p.Wo + TW - 1) / TW) * WM;
}

static inline int hv_conv3x3_launch(const hv_conv3x3_params& p, hipStream_t stream) {
    if (p.C1 <= 0 || p.C1 % 32 != 0 || p.C2 % 32 != 0 || p.Cout % 4 != 0) return -1;
    if (p.C2 > 0 && p.X2 == nullptr) return -1;
    if ((long)p.Cout * 9 * (p.C1 + p.C2) * 2 >= (1L << 32)) return -1;  // 32-bit weight offsets in the LDS-DMA path
    if (p.mode == HV_CONV_S1 && (p.Ho != p.Hs || p.Wo != p.Ws)) return -1;
    if (p.mode == HV_CONV_S2 && (p.Ho != (p.Hs + 1) / 2 || p.Wo != (p.Ws + 1) / 2)) return -1;
    if (p.mode == HV_CONV_UP2 && (p.Ho != 2 * p.Hs || p.Wo != 2 * p.Ws)) return -1;
    if (hv_conv_w4_applies(p)) {
        hv_conv_w4_launch(p, g_hv_conv_raster == 2 ? ((long)p.C1 * p.Cout >= 640L * 640 ? 1 : 0) : g_hv_conv_raster, stream);
        return 0;
    }
    // narrow images use the tall 16x8 patch so that the patch is not mostly padding; images with at
    // least 16 output rows use the 256-pixel (16x16) patch, except for the stride-2 form whose input
    // halo (33x33 pixels) would not fit two buffers in LDS
    const bool narrow = p.Wo <= 8;
    // (measured on MI355X: the 8-wave tile wins for the upsample-folded conv, 0.90 -> 1.03 PF/s, and
    //  loses for stride 1, where the per-step barrier over 8 waves costs more than the traffic saved; re-measured in round 3
    //  with the conflict-free LDS pitches, on 8 and on 4 waves: still no gain, profiles/r03_conv_tiles_ab.txt -- dispatch removed)
    const bool big = g_hv_conv_big && !narrow && p.Ho >= 16 && p.mode == HV_CONV_UP2;
    // 64-channel reduction chunks (whole 128-byte weight lines by LDS-DMA, 32 MFMAs per tap step): stride 1, both sources in
    // whole 64-channel chunks.  Hardware A/B (profiles/r03_hwcheck.txt, 48 images): 24x16 1280 -> 1280 0.746 -> 0.719 ms,
    // 12x8 0.258 -> 0.221 ms, but 96x64 320 -> 320 0.863 -> 0.928 and 48x32 640 -> 640 0.742 -> 0.770: the default takes
    // them for images of at most 384 output pixels (tuning value 3 forces them everywhere, 0 / 2 never).
    const bool ck64 = p.mode == HV_CONV_S1 && p.C1 % 64 == 0 && p.C2 % 64 == 0 &&
                      (g_hv_conv_big == 3 || (g_hv_conv_big == 1 && p.Ho * p.Wo <= 384));
    switch (p.mode) {
        case HV_CONV_S1:
            if (ck64 && narrow) hv_conv3x3_launch_t<8, HV_CONV_S1, 128, 64, 64>(p, stream);
            else if (ck64) hv_conv3x3_launch_t<16, HV_CONV_S1, 128, 64, 64>(p, stream);
            else if (narrow) hv_conv3x3_launch_t<8, HV_CONV_S1, 128>(p, stream);
            else hv_conv3x3_launch_t<16, HV_CONV_S1, 128>(p, stream);
            break;
        case HV_CONV_S2:
            narrow ? hv_conv3x3_launch_t<8, HV_CONV_S2, 128>(p, stream) : hv_conv3x3_launch_t<16, HV_CONV_S2, 128>(p, stream);
            break;
        case HV_CONV_UP2:
            if (big) hv_conv3x3_launch_t<16, HV_CONV_UP2, 256>(p, stream);
            else if (narrow) hv_conv3x3_launch_t<8, HV_CONV_UP2, 128>(p, stream);
            else hv_conv3x3_launch_t<16, HV_CONV_UP2, 128>(p, stream);
            break;
        default:
            return -1;
    }
    return 0;
}
