// hv_conv4.h -- stride-1 3x3 convolution on FOUR waves per CU (one per SIMD), the k-loop structure of hv_gemm_w4_kernel
// (hv_gemm4.h) applied to the implicit GEMM of hv_conv.h.  Same operation and parameter block as hv_conv3x3_kernel
// (reference: InflatedConv3d 3x3, /root/reference/src/models/resnet.py:9-15, with the time-embedding / residual adds of
// ResnetBlock3D, resnet.py:224-229, 243, in the epilogue); selected by hv_conv3x3_launch for stride-1 (plain or upsample-folded) single-source inputs without a
// GroupNorm prologue (the ResnetBlock3D convolutions of the denoising path read an activation that hv_affine_apply has
// normalised and concatenated) whose output channels come in blocks of 320.
//
// Why another structure.  hv_conv3x3_kernel gives a wave 64 pixels x 64 channels: 16 MFMAs per 8 ds_read_b128 and per
// barrier (32-channel chunks), three workgroups per CU -- per CU and tap step 768 LDS cycles of fragment reads beside 768
// matrix cycles per SIMD: the LDS pipe is as busy as the matrix pipe and the kernel sits at 0.38 - 0.42 of the MFMA peak
// (profiles/r05_final_kernel_stats.csv).  Here a wave owns 96 pixels x 160 channels (6 x 10 fragments, 240 accumulator
// registers): 60 MFMAs per 16 fragment reads, and a k-tile -- one tap of one 64-channel chunk -- carries 120 MFMAs per wave
// (1 920 matrix cycles) behind ONE barrier.  The convolution has what the four-wave GEMM lacked at K = 320: a deep reduction
// (9 Cin / 64 = 45 ... 360 k-tiles per tile), so the exposed prologue / epilogue of a one-wave-per-SIMD kernel is a few percent.
//
// Tile: 12 x 16 output pixels x 320 output channels per workgroup; waves 2 (pixel rows 0-5 / 6-11) x 2 (channels 0-159 /
// 160-319).  LDS (144 KiB): two halo buffers of 14 x 18 pixels x 64 channels (128-byte pixels, 16-byte pieces XOR-swizzled by
// the even key pixel & 6 on the source side -- a fragment's 16 consecutive pixels are conflict-free at any start: see the
// fragment addresses) and two weight slots of 320 rows x 128 bytes.  Everything reaches LDS by LDS-DMA:
//   * the weight tile of a k-tile: 40 wave-instructions of 1 KiB (ten per wave), rows in the order that gives a lane eight
//     CONSECUTIVE output channels over a fragment pair (LDS row 16 g + r <-> channel 32 (g >> 1) + 8 (r >> 2) + 4 (g & 1) + (r & 3)):
//     16-byte stores / residual loads in the epilogue;
//   * the halo of the NEXT 64-channel chunk: 32 wave-instructions, one per wave in each of the taps 0-7 of the current chunk.
//     Zero padding: the halo pixels outside the image are zeroed ONCE in both buffers and the copies run with those lanes
//     switched off in EXEC (the per-instruction masks are eight SGPR pairs) -- nothing ever writes there.
// k-tile s (tap of chunk), 20 blocks of 6 MFMAs (one weight fragment x the six pixel fragments):
//   blocks 0-16  |  vmcnt(0), lgkmcnt(0), barrier s + 1  |  first fragment reads of k-tile s + 1  |  blocks 17-19
// Weight fragments ride a 5-register ring (fragment b + 5 is requested right behind block b; none in blocks 15 / 16, so the
// barrier's lgkmcnt(0) waits for nothing), the pixel fragments of the second k half are requested in blocks 1-6, those of the
// next k-tile's first half behind the barrier: three blocks (288 matrix cycles) ahead of their use.  Behind barrier s + 1 every
// wave is done with k-tile s's slot: W(s + 2) goes there (two pieces at once, eight in blocks 1-8 of k-tile s + 1 -- at least
// eight blocks before the wait).  The loop has no conditional code: the copies past the end are clamped to the last chunk and
// land in slots nobody reads; the kernel drains them before it ends.
#pragma once
#include "hv_common.h"
#include "hv_gemm4.h"  // hv_glds16_u, hv_glds16_um, hv_lane_mask, hv_acc_take, hv_acc_settle, hv_mfma_tied
#include "humanvid_hip.h"

// MODE: HV_CONV_S1 (stride 1) or HV_CONV_UP2 (nearest-2x upsampling folded into the addressing, Upsample3D + conv,
// /root/reference/src/models/resnet.py:51-88: output pixel (oy, ox), tap (dy, dx) reads source pixel ((oy + dy - 1) >> 1,
// (ox + dx - 1) >> 1) -- the halo of a 12 x 16 output patch is 8 x 10 source pixels, ten LDS-DMA instructions per chunk)
// NF: weight fragments per wave = 10 (tile of 320 channels, 240 accumulator registers) or 8 (256 channels, 192 registers: where
// 320-wide tiles fill the 256 CUs badly -- Cout = 1280 at 24 x 16: 384 tiles = 1.5 rounds, 480 tiles of 256 = 1.875)
template <int MODE = HV_CONV_S1, int NF = 10>
#ifdef HV_C4_OLDKEY  // A/B build: the GEMM's swizzle key on the halo pixels (2-way conflicts on odd pixel pairs)
#define HV_C4_KEY(hp) (((hp) >> 1) & 7)
#else
#define HV_C4_KEY(hp) ((hp) & 6)
#endif
struct HvConv4Geom {
    static constexpr int TW = 16, TH = 12;
    static constexpr int HW = MODE == HV_CONV_S1 ? TW + 2 : TW / 2 + 2, HH = MODE == HV_CONV_S1 ? TH + 2 : TH / 2 + 2;
    static constexpr int HP = HW * HH;                       // 252 / 80 halo pixels
    static constexpr int NHJ = ((HP + 7) / 8 + 3) / 4;       // halo instructions per wave and chunk: 8 / 3
    static constexpr int BN = 32 * NF, WM = 2;
    static constexpr int RING = NF / 2, NB = 2 * NF;         // weight-fragment register ring (NB % RING == 0), blocks per k-tile
    static constexpr int HALO_B = 32768, WSLOT_B = BN * 128;
    static constexpr int W0 = 2 * HALO_B, W1 = W0 + WSLOT_B, LDS_B = W1 + WSLOT_B;  // 147 456 bytes
};

#ifdef HV_C4_TRACE
// timing build (tools/build_variant.sh c4trace k_conv -DHV_C4_TRACE): per workgroup (first 2048), wave 0: s_memtime at kernel
// start [0], behind barrier 0 [1], behind the k-loop [2], behind the epilogue's operand loads [3], at the end [4]
__device__ unsigned long long g_hv_c4_trace[2048 * 8];
#define HV_C4_MARK(i) if (tid == 0 && blockIdx.x < 2048) g_hv_c4_trace[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define HV_C4_MARK(i)
#endif

template <int MODE = HV_CONV_S1, int NF = 10>
__global__ __launch_bounds__(256, 1) void hv_conv_w4_kernel(hv_conv3x3_params p, int raster) {
    using G = HvConv4Geom<MODE, NF>;
    static_assert(NF == 10 || NF == 8, "tiles of 320 or 256 channels");
    constexpr int RING = G::RING, NB = G::NB, NJ = NF / 2;  // NJ: fragment pairs per wave = 16-byte channel groups per lane
    static_assert(MODE == HV_CONV_S1 || MODE == HV_CONV_UP2, "stride 1, plain or upsample-folded");
    constexpr bool UP = MODE == HV_CONV_UP2;
    constexpr int NHJ = G::NHJ;
    constexpr int TW = G::TW, TH = G::TH, HW = G::HW;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[G::LDS_B];

    const int tid = threadIdx.x, lane = tid & 63;
    HV_C4_MARK(0)
#ifndef HV_EMU
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#else
    const int wave = tid >> 6;
#endif
    const int wm = wave & 1, wn = wave >> 1, r16 = lane & 15, quad = lane >> 4;
    const int Cin = p.C1;

    // ---- tile of this workgroup (the raster of hv_conv3x3_kernel: XCD x owns a contiguous range of tiles)
    const int tiles_n = p.Cout / G::BN;
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    const int total = p.n_images * tiles_y * tiles_x * tiles_n;
    const int cpx = gridDim.x / 8;
    int t = (blockIdx.x % 8) * cpx + blockIdx.x / 8;
    if (t >= total) return;
    int n0;
    if (raster == 0) {
        n0 = (t % tiles_n) * G::BN;
        t /= tiles_n;
    } else {
        const int npatch = p.n_images * tiles_y * tiles_x;
        n0 = (t / npatch) * G::BN;
        t %= npatch;
    }
    const int x0 = (t % tiles_x) * TW;
    t /= tiles_x;
    const int y0 = (t % tiles_y) * TH;
    const int img = t / tiles_y;
    const int nchunks = Cin / 64;

    // ---- LDS-DMA sources.  Halo: instruction i = wave + 4 j (j = 0..7) carries halo pixels 8 i .. 8 i + 7, lane l the
    // 16-byte piece (l & 7) ^ key of pixel 8 i + (l >> 3).
    const char* const xsrc = reinterpret_cast<const char*>(p.X) + (long)img * p.Hs * p.Ws * Cin * 2;
    unsigned hofs[NHJ];
    unsigned long hmask[NHJ];
#pragma unroll
    for (int j = 0; j < NHJ; ++j) {
        const int i = wave + 4 * j;
        const int hp = 8 * i + (lane >> 3);
        const int row = hp / HW, col = hp - row * HW;
        const int iy = (UP ? y0 / 2 : y0) - 1 + row, ix = (UP ? x0 / 2 : x0) - 1 + col;
        const bool inb = hp < G::HP && iy >= 0 && iy < p.Hs && ix >= 0 && ix < p.Ws;
        const int piece = (lane & 7) ^ HV_C4_KEY(hp);  // halo swizzle key: see the fragment addresses below
        hofs[j] = inb ? (unsigned)(((iy * p.Ws + ix) * Cin + piece * 8) * 2) : 0u;
        hmask[j] = hv_lane_mask(inb);
        if (!inb) {  // zero padding, written once: the copies never touch these pieces
            hv_st16(smem + i * 1024 + lane * 16, u32x4{0u, 0u, 0u, 0u});
            hv_st16(smem + G::HALO_B + i * 1024 + lane * 16, u32x4{0u, 0u, 0u, 0u});
        }
    }
    // Weights: instruction i = wave + 4 j (j = 0..9) carries LDS rows 8 i .. 8 i + 7 = fragment g = (wave >> 1) + 2 j, rows
    // r = 8 (wave & 1) + (l >> 3); channel of (g, r) = 32 (g >> 1) + 4 (g & 1) + 8 (r >> 2) + (r & 3) = 32 j + [lane part].
    const char* const wsrc = reinterpret_cast<const char*>(p.W);
    unsigned wofs;
    {
        const int rp = 8 * (wave & 1) + (lane >> 3);
        const int chl = 4 * (wave >> 1) + 8 * (rp >> 2) + (rp & 3);
        const int key = (4 * (wave & 1) + (lane >> 4)) & 7;
        wofs = ((unsigned)(n0 + chl) * 9u * (unsigned)Cin + (unsigned)(((lane & 7) ^ key) * 8)) * 2u;
    }
    const long wj_stride = 32L * 9 * Cin * 2;  // bytes between the rows of consecutive j
    // k-tile (chunk c, tap) with c clamped to the last chunk (copies past the end of the reduction land in dead slots)
    auto issue_w = [&](int j, int c, int tap, unsigned slot_ofs) __attribute__((always_inline)) {
        const int cc = min(c, nchunks - 1);
        hv_glds16_u(wsrc + j * wj_stride + (long)(tap * Cin + cc * 64) * 2, wofs, smem + slot_ofs + (wave + 4 * j) * 1024);
    };
    auto issue_h = [&](int j, int c, unsigned buf_ofs) __attribute__((always_inline)) {
        const int cc = min(c, nchunks - 1);
        hv_glds16_um(xsrc + (long)cc * 128, hofs[j], smem + buf_ofs + (wave + 4 * j) * 1024, hmask[j]);
    };

    // ---- fragment addresses (LDS byte offsets).  Weights: row 160 wn + 16 nf + r16, piece (4 kk + quad) ^ ((r16 >> 1) & 7):
    // one lane offset + nf * 2048 (immediate) + the slot; the second k half is the first ^ 64.
    const unsigned wl = (unsigned)((16 * NF * wn + r16) * 128 + ((quad ^ ((r16 >> 1) & 7)) << 4));
    // Pixels: halo pixel (6 wm + s) * 18 + dx + r16 for s = mf + dy (0..7) and dx (0..2): 24 lane offsets, buffer included
    // (flipped by ^ HALO_B per chunk).  Swizzle key of a halo pixel: hp & 6 (EVEN keys only).  A ds_read_b128 is served in passes of
    // 16 lanes that mix two quads -- lanes r16 in {0-3, 12-15} of one quad (piece c) with r16 in {4-11} of its neighbour (piece
    // c ^ 1) -- over two 128-byte pixels per 256-byte bank row.  With the GEMM's key (hp >> 1) & 7 a window of 16 consecutive
    // pixels is conflict-free only when it starts at an even pixel PAIR (the GEMM's rows: multiples of 16); a tap window starts
    // anywhere, and the first version measured SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.20 - 0.27 (profiles/r06_lds_conflicts.txt
    // before this change).  Even keys never differ by 1, so the two quads' pieces cannot meet, and the four same-parity pixels of
    // each quad's lanes are four consecutive pairs -> four distinct keys: conflict-free for every start.
    unsigned xa[8][3];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int hp = UP ? ((6 * wm + s + 1) >> 1) * HW + ((r16 + dx + 1) >> 1) : (6 * wm + s) * HW + dx + r16;
            xa[s][dx] = (unsigned)(hp * 128 + ((quad ^ HV_C4_KEY(hp)) << 4));
        }

    f32x4 acc[NF][6];  // [nf][mf]
    bf16x8 wf[RING], xf[2][6];
    auto fence = [&]() __attribute__((always_inline)) {
#ifndef HV_EMU
        __builtin_amdgcn_sched_barrier(0);
#endif
    };
    // weight fragment g (0..19: k half g / 10, fragment g % 10) of the k-tile in the slot at `slot_ofs`
    auto rd_w = [&](unsigned slot_ofs, int g) __attribute__((always_inline)) {
        const unsigned a = (slot_ofs + wl) ^ (g >= NF ? 64u : 0u);
        return hv_as_bf16x8(hv_ld16(smem + a + (unsigned)(g % NF) * 2048u));
    };
    // pixel fragment mf of tap (dy, dx), k half kk; flip = HALO_B for the buffer of the next chunk
    auto rd_x = [&](int mf, int dy, int dx, int kk, unsigned flip) __attribute__((always_inline)) {
        return hv_as_bf16x8(hv_ld16(smem + (xa[mf + dy][dx] ^ flip ^ (kk ? 64u : 0u))));
    };

    // ---- prologue: halo of chunk 0, W(0); barrier 0; W(1) pieces 0, 1 and the first fragments of k-tile 0
    unsigned ws_even = G::W0, ws_odd = G::W1;  // slot of the k-tiles with even / odd tap in the current chunk (swapped per chunk: 9 taps)
    unsigned h_nxt = (unsigned)G::HALO_B;      // halo buffer of the NEXT chunk (flipped per chunk)
#pragma unroll
    for (int j = 0; j < NHJ; ++j) issue_h(j, 0, 0u);
#pragma unroll
    for (int j = 0; j < NF; ++j) issue_w(j, 0, 0, ws_even);
    // the accumulators are zeroed while the first copies are in flight (left to itself hipcc sinks the 4 NF x 6 writes behind
    // the barrier: ~1000 cycles per tile behind the wait instead of beside it)
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifndef HV_EMU
            asm volatile("" : "+a"(acc[a][b]));  // (pinned here)
#endif
        }
    hv_vm_wait<0>();
    hv_barrier_raw();
    HV_C4_MARK(1)
    issue_w(0, 0, 1, ws_odd);
    issue_w(1, 0, 1, ws_odd);
#pragma unroll
    for (int mf = 0; mf < 6; ++mf) xf[0][mf] = rd_x(mf, 0, 0, 0, 0u);
#pragma unroll
    for (int g = 0; g < RING; ++g) wf[g] = rd_w(ws_even, g);
    fence();

    int chunk = 0;
    do {  // (nchunks >= 1: a do-while leaves the accumulators ONE definition in front of the loop)
        hv_static_for<9>([&](auto T) __attribute__((always_inline)) {
            constexpr int tap = decltype(T)::value;
            constexpr int dy = tap / 3, dx = tap % 3;
            constexpr int tap1 = (tap + 1) % 9, tap2 = (tap + 2) % 9;  // the taps of k-tiles s + 1, s + 2
            constexpr int dy1 = tap1 / 3, dx1 = tap1 % 3;
            const unsigned ws_cur = (tap & 1) ? ws_odd : ws_even;  // slot of s and s + 2
            const unsigned ws_nxt = (tap & 1) ? ws_even : ws_odd;  // slot of s + 1
            const int c1 = chunk + (tap + 1) / 9, c2 = chunk + (tap + 2) / 9;
            constexpr unsigned flip1 = tap == 8 ? (unsigned)G::HALO_B : 0u;  // k-tile s + 1 reads the other halo buffer
            hv_static_for<NB>([&](auto B) __attribute__((always_inline)) {
                constexpr int b = decltype(B)::value, kk = b / NF, nf = b % NF;
                // copies: the halo piece of the next chunk in block 0 (taps 0-7), W(s + 1) pieces 2 .. NF-1 in blocks 1 .. NF-2
                if constexpr (b == 0 && tap < NHJ) issue_h(tap, chunk + 1, h_nxt);
                if constexpr (b >= 1 && b <= NF - 2) issue_w(b + 1, c1, tap1, ws_nxt);
                if constexpr (b == NB - 3) {
                    hv_vm_wait<0>();
                    hv_barrier_raw();  // lgkmcnt(0) + s_barrier: W(s + 1) (and at tap 8 the next halo) visible, slot of s free
                    issue_w(0, c2, tap2, ws_cur);
                    issue_w(1, c2, tap2, ws_cur);
#pragma unroll
                    for (int mf = 0; mf < 6; ++mf) xf[0][mf] = rd_x(mf, dy1, dx1, 0, flip1);
                    // the ring registers of the blocks NB - RING .. NB - 4 (no request behind them): the next k-tile's first fragments
#pragma unroll
                    for (int g = 0; g < RING - 3; ++g) wf[g] = rd_w(ws_nxt, g);
                }
#pragma unroll
                for (int mf = 0; mf < 6; ++mf)
                    hv_mfma_tied(acc[nf][mf], wf[b % RING], xf[kk][mf]);
                // requests behind the block: weight fragment b + RING (this k-tile, up to block NB - 1 - RING; the next one's behind
                // the three tail blocks), the second k half's pixel fragments in blocks 1-6
                if constexpr (b <= NB - 1 - RING) wf[b % RING] = rd_w(ws_cur, b + RING);
                if constexpr (b >= NB - 3) wf[b % RING] = rd_w(ws_nxt, b - (NB - RING));
                if constexpr (b >= 1 && b <= 6) xf[1][b - 1] = rd_x(b - 1, dy, dx, 1, 0u);
                fence();
            });
        });
        // nine taps per chunk: the slot parity of a tap flips, and so does the halo buffer
        const unsigned tmp = ws_even;
        ws_even = ws_odd;
        ws_odd = tmp;
        h_nxt ^= (unsigned)G::HALO_B;
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) xa[s][dx] ^= (unsigned)G::HALO_B;
    } while (++chunk < nchunks);
    hv_vm_wait<0>();  // the clamped copies past the end: landed before this workgroup's LDS is released
    HV_C4_MARK(2)

    // ---- epilogue: bias + time-embedding row + residual + activation, GroupNorm partial statistics of the stored values,
    // 16-byte stores (eight consecutive channels per lane and fragment pair j: channels n0 + 16 NF wn + 32 j + 8 quad ..).
    // ALL of the tile's operand loads are requested first -- one round trip (the first version asked for them fragment pair by
    // fragment pair: five dependent round trips, 17 - 24 us of a tile's 85, profiles/r06_s20_conv_w4_trace.txt) --, then the
    // arithmetic runs fragment pair by fragment pair and every result is stored where it is made (64-byte segments per
    // pixel), the store path busy beside the arithmetic.
    hv_acc_settle();
    const float* rv = p.rowvec ? p.rowvec + (long)(img / p.images_per_rowvec) * p.rowvec_ld : nullptr;
    const int rimg = p.residual ? (p.residual_images > 0 ? img % p.residual_images : img) : 0;
    const int gn_parts = tiles_y * tiles_x * G::WM;
    float* const gn_dst = p.gn_part ? p.gn_part + ((long)img * gn_parts + ((y0 / TH) * tiles_x + x0 / TW) * G::WM + wm) * p.Cout * 2
                                    : nullptr;
    const int nb = n0 + 16 * NF * wn + 8 * quad;  // + 32 j: the lane's eight channels of fragment pair j
    long opix[6];
#pragma unroll
    for (int mf = 0; mf < 6; ++mf) {
        const int oy = y0 + 6 * wm + mf, ox = x0 + r16;
        opix[mf] = (oy < p.Ho && ox < p.Wo) ? (long)(img * p.Ho + oy) * p.Wo + ox : -1;
    }
    f32x4 addv[NJ][2];
    u32x4 res[6][NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) a = *reinterpret_cast<const f32x4*>(p.bias + nb + 32 * j + 4 * h);
            if (rv) a += *reinterpret_cast<const f32x4*>(rv + nb + 32 * j + 4 * h);
            addv[j][h] = a;
        }
#pragma unroll
    for (int mf = 0; mf < 6; ++mf) {
        const long rpix = p.residual_images > 0 ? opix[mf] - (long)(img - rimg) * p.Ho * p.Wo : opix[mf];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            u32x4 r = {0u, 0u, 0u, 0u};
            if (p.residual && opix[mf] >= 0) r = hv_ld16(p.residual + rpix * p.Cout + nb + 32 * j);
            res[mf][j] = r;
        }
    }
    HV_C4_MARK(3)
    const bool want_stats = gn_dst != nullptr;  // (wave-uniform)
    auto finish = [&](auto ACT) __attribute__((always_inline)) {
        constexpr bool act = decltype(ACT)::value != 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float gs[8], gq[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) gs[e] = gq[e] = 0.f;
#pragma unroll
            for (int mf = 0; mf < 6; ++mf) {
                const f32x4 a0 = hv_acc_take(acc[2 * j][mf]) + addv[j][0], a1 = hv_acc_take(acc[2 * j + 1][mf]) + addv[j][1];
                float v[8], rf[8];
                hv_unpack8(res[mf][j], rf);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] = (e < 4 ? a0[e & 3] : a1[e & 3]) + rf[e];
                    if (act) v[e] = hv_act(v[e], p.out_act);
                }
                if (opix[mf] >= 0) {
                    hv_st16(p.Y + opix[mf] * p.Cout + nb + 32 * j, hv_pack8(v));
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        gs[e] += v[e];
                        gq[e] += v[e] * v[e];
                    }
                }
            }
            if (want_stats) {  // per channel: sum and sum of squares over this wave's valid pixels
                float a[8], b[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    a[e] = hv_row16_sum(gs[e]);
                    b[e] = hv_row16_sum(gq[e]);
                }
                if (r16 == 0) {  // channels n .. n + 7: {sum, sumsq} interleaved
#pragma unroll
                    for (int h = 0; h < 4; ++h)
                        *reinterpret_cast<f32x4*>(gn_dst + 2 * (nb + 32 * j) + 4 * h) = f32x4{a[2 * h], b[2 * h], a[2 * h + 1], b[2 * h + 1]};
                }
            }
        }
    };
    if (p.out_act == HV_ACT_NONE) finish(std::integral_constant<int, 0>{});
    else finish(std::integral_constant<int, 1>{});
    HV_C4_MARK(4)
}

// 0: never; 1: where the shape fills the tiles (default); 2: wherever the structure allows (tests); 3: as 1 with 320-channel tiles only (A/B)
static int g_hv_conv_w4 = 1;

// tile width of hv_conv_w4_kernel for this problem: 320 / 256 channels, or 0 = the kernel does not apply.  One workgroup per CU:
// the tiles must fill their rounds of 256 (>= 70 %, >= 128 tiles) and the 12 x 16 patches the image (>= 80 %); where both widths divide Cout the one whose
// tiles fill the rounds of 256 better wins (a 256-wide tile carries 96 instead of 120 MFMAs per wave behind each barrier: x 0.93).
static inline int hv_conv_w4_width(const hv_conv3x3_params& p) {
    if (g_hv_conv_w4 == 0) return 0;
    if ((p.mode != HV_CONV_S1 && p.mode != HV_CONV_UP2) || p.C2 != 0 || p.C1 <= 0 || p.C1 % 64 != 0 || p.Cout <= 0) return 0;
    if (p.Cout % 320 != 0 && p.Cout % 256 != 0) return 0;
    if (p.pro_scale != nullptr || p.pro_act != HV_ACT_NONE) return 0;  // the halo goes HBM -> LDS untouched
    if ((long)p.Hs * p.Ws * p.C1 * 2 >= (1L << 32)) return 0;           // 32-bit halo offsets inside an image
    constexpr int TH = HvConv4Geom<>::TH, TW = HvConv4Geom<>::TW;
    const long ty = (p.Ho + TH - 1) / TH, tx = (p.Wo + TW - 1) / TW;
    const long patches = ty * tx * p.n_images;
    auto fill = [&](int bn) {
        if (p.Cout % bn != 0) return 0.0;
        const long tiles = patches * (p.Cout / bn);
        return (double)tiles / (double)(((tiles + 255) / 256) * 256) * (bn == 256 ? 0.93 : 1.0);
    };
    const int bn = (g_hv_conv_w4 != 3 && fill(256) > fill(320)) || p.Cout % 320 != 0 ? 256 : 320;  // (tuning 3: 320-wide only, A/B)
    if (p.Cout % bn != 0) return 0;
    if (g_hv_conv_w4 == 2) return bn;
    const double cover = (double)p.Ho * p.Wo / (double)(ty * TH * tx * TW);
    // one workgroup per CU: what counts is how well the tiles fill their rounds of 256.  (By tile count >= 384 -- the first rule --
    // the per-rank shapes of a sharded job fell back too early: 240 tiles are one round at 94 %.  Same-box steps at the per-rank
    // shapes, profiles/r06_s32_conv_w4_fill.txt: 3 frames 23.86 -> 23.43 ms, one CFG half at 6 frames 23.49 -> 23.14, N = 1 unchanged;
    // 1.5 rounds of 320-wide tiles at level 2 of config #3 measured equal to the 128-channel kernel: profiles/r06_s19_conv_w4.txt.)
    const long tiles = patches * (p.Cout / bn);
    const double f = (double)tiles / (double)(((tiles + 255) / 256) * 256);
    return cover >= 0.8 && tiles >= 128 && f >= 0.70 ? bn : 0;
}
static inline bool hv_conv_w4_applies(const hv_conv3x3_params& p) { return hv_conv_w4_width(p) != 0; }

static inline void hv_conv_w4_launch(const hv_conv3x3_params& p, int raster, hipStream_t stream) {
    constexpr int TH = HvConv4Geom<>::TH, TW = HvConv4Geom<>::TW;
    const int bn = hv_conv_w4_width(p);
    const int tiles = p.n_images * ((p.Ho + TH - 1) / TH) * ((p.Wo + TW - 1) / TW) * (p.Cout / bn);
    const int grid = ((tiles + 7) / 8) * 8;
    hv_note("hv_conv_w4_kernel<%s%d> | n=%d Hs=%d Ws=%d Ho=%d Wo=%d Cin=%d Cout=%d gn=%d res=%d", p.mode == HV_CONV_UP2 ? "up2," : "", bn,
            p.n_images, p.Hs, p.Ws, p.Ho, p.Wo, p.C1, p.Cout, 0, p.residual != nullptr);
    if (p.mode == HV_CONV_UP2) {
        if (bn == 256) hv_launch(hv_conv_w4_kernel<HV_CONV_UP2, 8>, dim3(grid), dim3(256), stream, p, raster);
        else hv_launch(hv_conv_w4_kernel<HV_CONV_UP2, 10>, dim3(grid), dim3(256), stream, p, raster);
    } else {
        if (bn == 256) hv_launch(hv_conv_w4_kernel<HV_CONV_S1, 8>, dim3(grid), dim3(256), stream, p, raster);
        else hv_launch(hv_conv_w4_kernel<HV_CONV_S1, 10>, dim3(grid), dim3(256), stream, p, raster);
    }
}
