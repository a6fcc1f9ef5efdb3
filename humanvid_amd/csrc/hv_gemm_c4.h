// hv_gemm_c4.h -- Y = X . W^T + bias (+ residual) on 192 x 320 x 64 tiles, four waves per CU (one per SIMD): the k-loop of
// hv_conv_w4_kernel (hv_conv4.h) with one "tap" -- for the deep-K projections whose N is a multiple of 320 and therefore fits
// neither the 256-wide tiles of hv_gemm_w4_kernel nor (N = 640) the 256 x 320 tiles of hv_gemm_wide_kernel: the feed-forward
// output projections of levels 0 and 1 (reference: FeedForward net[2] under BasicTransformerBlock / TemporalBasicTransformerBlock,
// /root/reference/src/models/attention.py:427; K = 4 C = 1280 / 2560, N = C = 320 / 640, residual in place).
//
// Waves 2 (rows 0-95 / 96-191) x 2 (channels 0-159 / 160-319): 6 x 10 fragments, 240 accumulator registers, 120 MFMAs per wave
// and k-tile behind ONE barrier, 16 fragment reads per 60 MFMAs.  LDS (152 KiB): X ring of three 192 x 64 slots (rows of 128
// bytes, pieces XOR-swizzled by (row >> 1) & 7 on the source side), W ring of two 320 x 64 slots, both filled by LDS-DMA:
//   k-tile s:  blocks 0-16  |  vmcnt(6), lgkmcnt(0), barrier s + 1  |  first fragment reads of s + 1  |  blocks 17-19
//   copies:    W(s + 1) pieces 0, 1 behind barrier s, 2-9 in blocks 0-7 of k-tile s (L2 hits: the tile's W panel is shared
//              by every workgroup), X(s + 2) pieces 0-5 in blocks 8-13 (HBM: two k-tiles ahead) -- the wait in front of barrier
//              s + 1 leaves exactly those six in flight.
// The weight rows are assigned as in hv_conv_w4_kernel (a lane holds eight consecutive channels per fragment pair): 16-byte
// residual loads and stores.  Accumulation runs over ascending k from a zero accumulator and the epilogue is operation for
// operation hv_gemm_epilogue_fast_perm<., false, RES>: outputs are bit-identical to every other kernel selection of hv_gemm.
// No statistics, tables, GEGLU or LayerNorm fold here: hv_gemm_choose sends only plain bias (+ residual) problems.
#pragma once
#include "hv_common.h"
#include "hv_gemm4.h"  // hv_glds16_u, hv_acc_take, hv_acc_settle, hv_mfma_tied
#include "humanvid_hip.h"

struct HvGemmC4Geom {
    static constexpr int BM = 192, BN = 320;
    static constexpr int XSLOT_B = BM * 128, WSLOT_B = BN * 128;          // 24 576 / 40 960 bytes
    static constexpr int W0 = 3 * XSLOT_B, LDS_B = W0 + 2 * WSLOT_B;      // 155 648 bytes
};

template <int V = 0>  // (a template only so that the header may be included by several translation units)
__global__ __launch_bounds__(256, 1) void hv_gemm_c4_kernel(hv_gemm_params p) {
    using G = HvGemmC4Geom;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[G::LDS_B];
    const int tid = threadIdx.x, lane = tid & 63;
#ifndef HV_EMU
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#else
    const int wave = tid >> 6;
#endif
    const int wm = wave & 1, wn = wave >> 1, r16 = lane & 15, quad = lane >> 4;

    // tile: XCD x owns a contiguous range of tiles; the column tiles of a row block are adjacent (its X is fetched once and
    // re-read from L2 by the other column tiles)
    const int tiles_n = p.N / G::BN, total = (p.M / G::BM) * tiles_n;
    const int cpx = gridDim.x / 8;
    const int t = (blockIdx.x % 8) * cpx + blockIdx.x / 8;
    if (t >= total) return;
    const int n0 = (t % tiles_n) * G::BN, m0 = (t / tiles_n) * G::BM;
    const int nk = p.K / 64;

    // ---- LDS-DMA sources.  X: instruction i = wave + 4 j (j = 0..5) carries rows 8 i .. 8 i + 7 of the tile, lane l the
    // 16-byte piece (l & 7) ^ key of row 8 i + (l >> 3); key = (row >> 1) & 7 = (4 (wave & 1) + (l >> 4)) & 7 for every j.
    const int key = (4 * (wave & 1) + (lane >> 4)) & 7;
    const char* const xsrc = reinterpret_cast<const char*>(p.X) + (long)(m0 + 8 * wave) * p.ldx * 2;
    const unsigned xofs = ((unsigned)(lane >> 3) * (unsigned)p.ldx + (unsigned)(((lane & 7) ^ key) * 8)) * 2u;
    const long xj_stride = 32L * p.ldx * 2;
    // W: as hv_conv_w4_kernel -- LDS rows 8 i .. 8 i + 7 = fragment g = (wave >> 1) + 2 j, rows r = 8 (wave & 1) + (l >> 3);
    // channel of (g, r) = 32 (g >> 1) + 4 (g & 1) + 8 (r >> 2) + (r & 3) = 32 j + [lane part]
    const char* const wsrc = reinterpret_cast<const char*>(p.W);
    unsigned wofs;
    {
        const int rp = 8 * (wave & 1) + (lane >> 3);
        const int chl = 4 * (wave >> 1) + 8 * (rp >> 2) + (rp & 3);
        wofs = ((unsigned)(n0 + chl) * (unsigned)p.K + (unsigned)(((lane & 7) ^ key) * 8)) * 2u;
    }
    const long wj_stride = 32L * p.K * 2;
    // (k-tiles past the end are clamped to the last one: the copies land in slots nobody reads)
    auto issue_x = [&](int j, int s, unsigned slot_ofs) __attribute__((always_inline)) {
        hv_glds16_u(xsrc + j * xj_stride + (long)min(s, nk - 1) * 128, xofs, smem + slot_ofs + (wave + 4 * j) * 1024);
    };
    auto issue_w = [&](int j, int s, unsigned slot_ofs) __attribute__((always_inline)) {
        hv_glds16_u(wsrc + j * wj_stride + (long)min(s, nk - 1) * 128, wofs, smem + slot_ofs + (wave + 4 * j) * 1024);
    };

    // ---- fragment addresses inside a slot: row * 128 + ((4 kk + quad) ^ ((row >> 1) & 7)) * 16; the rows of a wave's fragments
    // are 16 apart (2048 bytes: an immediate), the second k half is ^ 64
    const unsigned wl = (unsigned)((160 * wn + r16) * 128 + ((quad ^ ((r16 >> 1) & 7)) << 4));
    const unsigned xl = (unsigned)((96 * wm + r16) * 128 + ((quad ^ ((r16 >> 1) & 7)) << 4));

    f32x4 acc[10][6];  // [nf][mf]
    bf16x8 wf[5], xf[2][6];
    auto fence = [&]() __attribute__((always_inline)) {
#ifndef HV_EMU
        __builtin_amdgcn_sched_barrier(0);
#endif
    };
    auto rd_w = [&](unsigned slot_ofs, int g) __attribute__((always_inline)) {
        const unsigned a = (slot_ofs + wl) ^ (g >= 10 ? 64u : 0u);
        return hv_as_bf16x8(hv_ld16(smem + a + (unsigned)(g % 10) * 2048u));
    };
    auto rd_x = [&](unsigned slot_ofs, int mf, int kk) __attribute__((always_inline)) {
        const unsigned a = (slot_ofs + xl) ^ (kk ? 64u : 0u);
        return hv_as_bf16x8(hv_ld16(smem + a + (unsigned)mf * 2048u));
    };

    // ---- prologue: X(0), X(1), W(0); barrier 0; W(1) pieces 0, 1 and the first fragments of k-tile 0
    unsigned xs_cur = 0u, xs_nxt = (unsigned)G::XSLOT_B, xs_aft = 2u * (unsigned)G::XSLOT_B;  // slots of X(s), X(s + 1), X(s + 2)
    unsigned ws_cur = (unsigned)G::W0, ws_nxt = (unsigned)(G::W0 + G::WSLOT_B);
#pragma unroll
    for (int j = 0; j < 6; ++j) issue_x(j, 0, xs_cur);
#pragma unroll
    for (int j = 0; j < 10; ++j) issue_w(j, 0, ws_cur);
#pragma unroll
    for (int j = 0; j < 6; ++j) issue_x(j, 1, xs_nxt);
    // (zeroed while the first copies are in flight: see hv_conv_w4_kernel)
#pragma unroll
    for (int a = 0; a < 10; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifndef HV_EMU
            asm volatile("" : "+a"(acc[a][b]));
#endif
        }
    hv_vm_wait<0>();
    hv_barrier_raw();
    issue_w(0, 1, ws_nxt);
    issue_w(1, 1, ws_nxt);
#pragma unroll
    for (int mf = 0; mf < 6; ++mf) xf[0][mf] = rd_x(xs_cur, mf, 0);
#pragma unroll
    for (int g = 0; g < 5; ++g) wf[g] = rd_w(ws_cur, g);
    fence();

    int s = 0;
    do {
        hv_static_for<20>([&](auto B) __attribute__((always_inline)) {
            constexpr int b = decltype(B)::value, kk = b / 10, nf = b % 10;
            if constexpr (b <= 7) issue_w(b + 2, s + 1, ws_nxt);
            if constexpr (b >= 8 && b <= 13) issue_x(b - 8, s + 2, xs_aft);
            if constexpr (b == 17) {
                hv_vm_wait<6>();   // W(s + 1) and X(s + 1) landed; the six pieces of X(s + 2) stay in flight
                hv_barrier_raw();  // lgkmcnt(0) + s_barrier: every wave is done with the slots of k-tile s
                issue_w(0, s + 2, ws_cur);
                issue_w(1, s + 2, ws_cur);
#pragma unroll
                for (int mf = 0; mf < 6; ++mf) xf[0][mf] = rd_x(xs_nxt, mf, 0);
                wf[0] = rd_w(ws_nxt, 0);
                wf[1] = rd_w(ws_nxt, 1);
            }
#pragma unroll
            for (int mf = 0; mf < 6; ++mf) hv_mfma_tied(acc[nf][mf], wf[b % 5], xf[kk][mf]);
            if constexpr (b <= 14) wf[b % 5] = rd_w(ws_cur, b + 5);
            if constexpr (b >= 17) wf[b % 5] = rd_w(ws_nxt, b - 15);
            if constexpr (b >= 1 && b <= 6) xf[1][b - 1] = rd_x(xs_cur, b - 1, 1);
            fence();
        });
        const unsigned x0 = xs_cur;
        xs_cur = xs_nxt, xs_nxt = xs_aft, xs_aft = x0;
        const unsigned w0 = ws_cur;
        ws_cur = ws_nxt, ws_nxt = w0;
    } while (++s < nk);
    hv_vm_wait<0>();  // the clamped copies past the end: landed before this workgroup's LDS is released

    // ---- epilogue: (acc + bias) + residual, 16-byte stores; all operand loads first (one round trip)
    hv_acc_settle();
    const int nb = n0 + 160 * wn + 8 * quad;  // + 32 j: the lane's eight channels of fragment pair j
    const int mb = m0 + 96 * wm + r16;        // + 16 mf
    f32x4 addv[5][2];
    u32x4 res[6][5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) a = *reinterpret_cast<const f32x4*>(p.bias + nb + 32 * j + 4 * h);
            addv[j][h] = a;
        }
#pragma unroll
    for (int mf = 0; mf < 6; ++mf)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            u32x4 r = {0u, 0u, 0u, 0u};
            if (p.residual) r = hv_ld16(p.residual + (long)(mb + 16 * mf) * p.ldr + nb + 32 * j);
            res[mf][j] = r;
        }
    const bool has_res = p.residual != nullptr;
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int mf = 0; mf < 6; ++mf) {
            f32x4 v0 = hv_acc_take(acc[2 * j][mf]) + addv[j][0], v1 = hv_acc_take(acc[2 * j + 1][mf]) + addv[j][1];
            if (has_res) {
                const u32x4 r = res[mf][j];
                v0 += f32x4{hv_bf2f((bf16_t)(r[0] & 0xffff)), hv_bf2f((bf16_t)(r[0] >> 16)), hv_bf2f((bf16_t)(r[1] & 0xffff)),
                            hv_bf2f((bf16_t)(r[1] >> 16))};
                v1 += f32x4{hv_bf2f((bf16_t)(r[2] & 0xffff)), hv_bf2f((bf16_t)(r[2] >> 16)), hv_bf2f((bf16_t)(r[3] & 0xffff)),
                            hv_bf2f((bf16_t)(r[3] >> 16))};
            }
            hv_st16(reinterpret_cast<uint16_t*>(p.Y) + (long)(mb + 16 * mf) * p.ldy + nb + 32 * j,
                    u32x4{hv_pack2(v0[0], v0[1]), hv_pack2(v0[2], v0[3]), hv_pack2(v1[0], v1[1]), hv_pack2(v1[2], v1[3])});
        }
}
