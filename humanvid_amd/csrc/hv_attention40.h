// hv_attention40.h -- spatial self-attention with reference-bank keys for head dim 40 (SD-1.5 level 0: 320 channels over
// 8 heads), the shape that carries 19 % of a denoising step at 24f x 768x512 (five launches of 48 images x 6144 queries
// against 6144 own + 6144 bank keys).  Same semantics and parameter block as hv_attention (hv_attention.h; reference:
// /root/reference/src/models/mutual_self_attention.py:147-186 over diffusers' AttnProcessor2_0), other structure.
//
// What the round-3 measurements said about the generic kernel at this shape (profiles/r03_attn_ablation.txt, 4.2 ms per
// launch): the MFMAs alone take 1.47 ms -- 40 % of them padding (QK^T 40 -> 64 deep, PV 40 -> 48 rows) --, the 14 fragment
// reads per tile and wave cost 1.5 ms, staging the K / V^T tiles through registers into LDS 1.55 ms, softmax VALU 1.1 ms.
// This kernel (round 4) changes the four things those numbers point at:
//  * S^T = K.Q^T on v_mfma_f32_32x32x16_bf16: 16-deep steps pad the head dim to 48 instead of 64 (-25 % QK^T MFMA time),
//    one K fragment read feeds a 32-cycle MFMA over 32 queries (six K reads per 64-key tile and wave instead of eight), and
//    no accumulation chain mixes MFMA shapes (the gfx950 hazard of profiles/r02_mfma_chain_hazard.md): S^T chains are
//    32x32x16 only, O^T chains 16x16x32 only;
//  * the probabilities go from the 32 x 32 S^T layout (lane = query l & 31, half l >> 5 holds 16 of the block's 32 keys)
//    to the B operand of O^T += V^T.P^T (16x16x32: lane = query l & 15, quad l >> 4 holds 8 keys) with four
//    v_permlane16_swap per 32-key block -- the K rows are staged in the order that makes the swapped dwords line up with
//    the natural key order of V^T (below), so nothing else moves;
//  * 256 queries per workgroup (8 waves x 32): a staged K / V^T tile serves twice the queries of the generic kernel's 128
//    -- half the staging and barriers per query; two workgroups per CU;
//  * tiles staged by LDS-DMA (ten global_load_lds_dwordx4 wave-instructions per 64-key tile, inline asm so that hipcc's
//    wait-count tracker does not drain them in front of every fragment read): no staging registers, no LDS stores, no LDS
//    initialisation, 20 KB of LDS.  The constants the MFMAs want beside the data -- the augmented K column, the ones / zero
//    rows of V^T -- are three 16-byte chunks behind each buffer that the lanes concerned address instead of tile data
//    (HvAttn40Geom).  The register-staged form it replaces (ds_write_b128 of a chunk per thread, padded pitches) measured
//    1 % slower (profiles/r04_s17_attn40_dma.txt): staging is not what this kernel waits for either;
//  * softmax VALU work per score: the query's reference maximum rides in the MFMA (K is augmented by a column of ones at
//    head-dim index 40, Q by -m there: the MFMA delivers s - m with C = 0 -- no accumulator initialisation moves, no
//    subtraction), the "some probability exceeds 2^THR" test runs on the PACKED bf16 pairs (v_pk_max_u16: positive bf16
//    order like unsigned integers -- 15 instead of 31 maxima), exponentials in the exp2 domain as before.
//  * OPTIMISTIC reference maximum: the first tile's maximum is kept as the reference for the whole key loop and NO per-tile
//    test runs (the "some probability exceeds 2^THR" test of the generic kernel -- 15 packed maxima, a compare and a
//    wave vote per tile -- measured 6.3 % of the kernel, profiles/r04_s3.txt).  bf16 probabilities and fp32 accumulators
//    keep their relative precision at any magnitude, so nothing is lost until a later score exceeds the reference by ~2^7
//    in the exp2 domain and the exponential overflows; the denominator row of O^T then is not finite, the workgroup votes
//    once after the loop, and -- only then -- runs the loop again in the careful form (test + deferred rescale per tile).
//  (Round 4, measured and removed -- commit dd5f417 has the code: a TWO-GROUP loop, waves 0-3 and 4-7 half a tile apart so that one
//   group's MFMA phase (PV of tile t, S of tile t + 1) runs beside the other's VALU phase on every SIMD, a workgroup barrier per
//   phase, three tile buffers, the careful form as a non-inlined function.  Bit-identical results, 127 registers, but 3-5 %
//   SLOWER than this single-phase loop on the same box (profiles/r04_s10_attn40_two_group.txt): a phase lasts ~1460 cycles for
//   768 matrix-pipe and ~830 VALU cycles of work, the rest is the second barrier per tile and staging.  The issue-rate
//   measurements behind the attempt (tools/valu_rate.hip, profiles/r04_s9_issue_rates.txt): v_exp_f32 8.4 cycles per
//   wave-instruction (10.6-11.4 beside another wave's MFMA stream, which is not slowed), v_cvt_pk_bf16_f32 4.8,
//   v_permlane16_swap 8.6, plain VALU 2.5; per 64-key tile and wave that is ~410-540 VALU cycles against 384 matrix-pipe
//   cycles -- this kernel is bound by its softmax VALU work, and 778 cycles per wave-tile and SIMD are what it takes now.)
//  (What bounds it, profiles/r04_s14_attn40_stream_model.txt: a synthetic wave with exactly this kernel's per-tile instruction
//   stream and no memory side -- 6 + 12 MFMAs, 32 v_exp_f32, 16 packs, 8 lane-row swaps, dependent as here -- runs at 603-609
//   cycles per wave-tile and SIMD with four waves per SIMD (its VALU part alone: 530; its MFMAs alone: 384), this kernel at
//   ~780: 78 % of what the arithmetic alone allows, the rest is the tile barrier, LDS latency and staging.  One workgroup per CU
//   instead of two: 4.91 instead of 3.85 ms.  Truncating instead of rounding the probabilities (v_perm_b32, half the issue
//   time of the pack): no change, reverted.)
//  (Wave priority: raising the MFMA sections (s_setprio 1, the generic kernel's -1.7 %), flat, raising the softmax section:
//   3.845 / 3.802 / 3.910 ms on one box, profiles/r04_s13_attn40_prio.txt -- flat it is.)
// The reference maximum therefore lives in bf16 (it is an element of the Q operand); the rescale factors are computed from
// the rounded values, so every tile of a query is exponentiated against exactly the maximum its O^T / denominator carry
// (the softmax is invariant to the choice of reference as long as it is used consistently).
//
// Key order.  S^T block (32 keys x 32 queries), C/D layout of the 32x32 MFMA: lane l, register i <-> row (i & 3) + 8 (i >> 2)
// + 4 (l >> 5).  K tile rows are staged (source-side, by the DMA's per-lane addresses) so that LDS row 8 a + 4 b + c of a
// 32-key block holds key 16 b + 4 a + c: register i
// of half b then IS key 16 b + i, the packed pair j (registers 2 j, 2 j + 1) keys 16 b + 2 j (+1), and after
// v_permlane16_swap(pair d, pair d + 4) quad q of the result holds keys 8 q + 2 d (+1) for the queries r16 (first result:
// queries 0-15 of the wave's 32, second: 16-31) -- dword d of the PV B operand with V^T in natural key order.
#pragma once
#include "hv_common.h"
#include "humanvid_hip.h"

#ifndef HV_ATTN_THR
#define HV_ATTN_THR 8.0f
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short hv_u16x2 __attribute__((ext_vector_type(2)));

// v_permlane16_swap: odd 16-lane rows of `a` are exchanged with even rows of `b`
HV_DEV void hv_swap16_pair(unsigned& a, unsigned& b) {
#ifndef HV_EMU
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
#else
    const int lane = threadIdx.x & 63;
    const bool odd = (lane >> 4) & 1;
    const unsigned xa = __shfl(b, lane ^ 16), xb = __shfl(a, lane ^ 16);
    a = odd ? xa : a;
    b = odd ? b : xb;
#endif
}
HV_DEV unsigned hv_pk_max_u16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(hv_u16x2, a), __builtin_bit_cast(hv_u16x2, b)));
}

struct HvAttn40Geom {
    static constexpr int D = 40, NW = 8, BQ = 32 * NW;
    // Tiles are staged by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes land contiguously), so a tile in LDS is a dense
    // array of 16-byte chunks and every constant the MFMAs need beside the data sits in a few chunks BEHIND the tile that the
    // lanes concerned address instead of tile data:
    //   K tile: 64 rows x 80 bytes (5 chunks: 320 chunks = five wave-instructions).  20 dwords per row: the 16 rows of a
    //     ds_read_b128 lane group fall on 16 different 4-dword bank windows (20 r mod 64 over the group's rows is a permutation
    //     of the multiples of 4).  The augmented column (bf16 1.0 at head-dim index 40, zeros up to 47) is one constant chunk:
    //     the upper-half lanes of the third k-step read it for every key row (a broadcast).
    //   V^T tile: 40 rows x 128 bytes (8 chunks of 8 keys: 320 chunks), chunk position XOR-swizzled by the row (position c of
    //     row r holds keys 8 (c ^ (r & 7)) .. + 7, applied on the source side of the DMA): conflict-free for the 16x16x32
    //     A-operand read.  Row 40 (all ones: the denominator) and rows 41 - 47 (zeros) are two constant chunks.
    //     Round 5: rows 40 - 47 are laid out like tile rows -- 128 bytes of ones (row 40) and two 128-byte rows of zeros (the
    //     odd rows 41 .. 47 read row 41, the even rows 42 .. 46 row 42, each at its own swizzled chunk position) -- so that
    //     the lanes of a ds_read_b128 group that read constants hit the 16-byte windows the design reserves for rows 40 - 47
    //     instead of all landing on one chunk next to the data lanes': 8 extra LDS cycles per 24 group reads before, none now
    //     (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.14 in round 4, profiles/r04_lds_conflicts.txt).
    static constexpr int KRS = 80, VRS = 128;
    static constexpr int KBYTES = 64 * KRS, VBYTES = D * VRS;     // 5120 each
    static constexpr int KSTRIDE = KBYTES + 16, VSTRIDE = VBYTES + 384;  // buffer strides: tile + its constant chunks / rows
    static constexpr int NBUF = 2;  // (round 5, measured and not kept: four buffers and ONE barrier per two tiles -- 1-2 % slower on the
                                    //  same box, profiles/r05_s5_attn40.txt: the barrier is not what the tile waits for)
    static constexpr int LDS_BYTES = NBUF * KSTRIDE + NBUF * VSTRIDE;
};

template <bool MASK>
__global__ __launch_bounds__(512, 4) void hv_attention40_kernel(hv_attention_params p) {
    using G = HvAttn40Geom;
    constexpr int D = G::D;
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::LDS_BYTES];
    unsigned char* Ks = smem;                   // buffer b: Ks + b * KSTRIDE  (tile, then the augmented-column chunk)
    unsigned char* Vs = smem + G::NBUF * G::KSTRIDE;  // buffer b: Vs + b * VSTRIDE  (tile, then the ones row 40, then the zeros row)

    const int tid = threadIdx.x, lane = tid & 63;
#ifndef HV_EMU
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#else
    const int wave = tid >> 6;
#endif
    const int r16 = lane & 15, quad = lane >> 4, l32 = lane & 31, half = lane >> 5;

    const int nqb = (p.Lq + G::BQ - 1) / G::BQ;
    const int total = nqb * p.heads * p.n_images;
    const int cpx = gridDim.x / 8;
    int t = (blockIdx.x % 8) * cpx + blockIdx.x / 8;  // XCD x walks the contiguous range [x cpx, (x + 1) cpx)
    if (t >= total) return;
    // the query blocks of an (image, head) run next to each other: its K / V^T stream through L2 once.  (The other raster --
    // the 8 heads of a query block adjacent, so that their 80-byte output pieces fill whole lines in L2 -- measured 2 % slower
    // on MI355X, profiles/r04_s1.txt: this kernel moves 0.45 TB/s, it is not bound by its HBM bytes.)
    const int qb = t % nqb;
    t /= nqb;
    const int head = t % p.heads;
    int img = t / p.heads;
    // alternate between the CFG halves (the conditional images attend to twice the keys): every XCD gets the same mix -- and
    // the LONG half first in every pair, so that an XCD's range ends with short workgroups (its 64 slots are refilled in
    // order: the tail of the launch is one short workgroup instead of one long one; profiles/r06_s30_attn40_order.txt)
    if ((p.n_images & 1) == 0) {
#ifndef HV_ATTN40_ORDER_OLD
        const bool long_second = p.bank_sel != nullptr && p.L2 > 0 && p.bank_sel[p.n_images >> 1] >= 0 && p.bank_sel[0] < 0;
        img = ((img & 1) ^ (long_second ? 1 : 0)) * (p.n_images >> 1) + (img >> 1);
#else
        img = (img & 1) * (p.n_images >> 1) + (img >> 1);
#endif
    }
    const int sel = (p.bank_sel != nullptr && p.L2 > 0) ? p.bank_sel[img] : -1;
    const int T1 = (p.L1 + 63) / 64;
    const int T2 = sel >= 0 ? (p.L2 + 63) / 64 : 0;
    const int ntiles = T1 + T2;

    // LDS constants (every tile byte is written by the DMA; the first tile barrier orders these stores before their readers)
    if (tid < G::NBUF) hv_st16(Ks + tid * G::KSTRIDE + G::KBYTES, u32x4{0x00003F80u, 0u, 0u, 0u});
    if (tid >= 64 && tid < 64 + 24 * G::NBUF) {  // 8 chunks of ones + 16 chunks of zeros behind every V^T buffer
        const int i = tid - 64, b = i / 24, c = i - 24 * b;
        const unsigned one2 = c < 8 ? 0x3F803F80u : 0u;
        hv_st16(Vs + b * G::VSTRIDE + G::VBYTES + 16 * c, u32x4{one2, one2, one2, one2});
    }

    // ---- query fragments: B operand of S^T = K.Q^T (32x32x16): lane = query l32, elements = head-dim 16 s + 8 half + 0..7;
    //      pre-multiplied by scale * log2(e) (scores come out of the MFMA in the exp2 domain); step 2 of the upper half is the
    //      augmented part: element 0 carries -m (the query's reference maximum, negated), the rest is zero
    bf16x8 qf[3];
    const int q_wave = qb * G::BQ + wave * 32;
    {
        const int q = q_wave + l32;
        const bf16_t* qrow = p.Q + ((long)img * p.Lq + q) * p.ldq + head * D;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (q < p.Lq && 16 * s + 8 * half + 8 <= D) v = hv_ld16(qrow + 16 * s + 8 * half);
            float f[8];
            hv_unpack8(v, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] *= p.scale * 1.44269504089f;
            qf[s] = hv_as_bf16x8(hv_pack8(f));
        }
    }
    float mneg = 0.f;  // -m of this lane's query: always exactly representable in bf16 (it is what qf[2][0] of the upper half holds)

    // ---- tile staging by LDS-DMA: waves 0-4 issue one instruction of the K tile each, waves 3-7 one of the V^T tile (source =
    //      wave-uniform tile base + a per-lane byte offset that does not change from tile to tile), into the buffer the
    //      previous tile was read from two barriers ago.  No staging registers, no LDS stores, no LDS initialisation.
    const bool k_loader = wave < 5, v_loader = wave >= 3;  // (wave-uniform)
    // K: chunk 64 wave + lane of the tile = row (chunk / 5), 16-byte piece (chunk % 5) of its 80-byte head slice; LDS row
    //    8 a + 4 b + c of a 32-key block holds key 16 b + 4 a + c (see the header)
    const int kch = 64 * (k_loader ? wave : 0) + lane, klrow = kch / 5, kpc = kch - 5 * klrow;
    const int kkey = (klrow & 32) | (((klrow >> 2) & 1) << 4) | (((klrow >> 3) & 3) << 2) | (klrow & 3);
    // V^T: chunk 64 (wave - 3) + lane = channel row (chunk / 8), position (chunk % 8) <- keys 8 (position ^ (row & 7)) .. + 7
    const int vch = 64 * (v_loader ? wave - 3 : 0) + lane, vrow = vch >> 3, vkc = (vch & 7) ^ (vrow & 7);
    unsigned koff = 0, voff = 0;      // per-lane source byte offsets within a tile (set_source)
    const char* kbase = nullptr;      // wave-uniform: first key row of the next tile to request, at this head's columns
    const char* vbase = nullptr;      //               first key column of the next tile, at this head's first channel row
    unsigned kstep = 0, ldk_b = 0, ldv_b = 0;
    int src_kv0 = 0, src_L = 0;
    auto set_source = [&](bool bank) {
        const unsigned rowbase = bank ? (unsigned)sel * (unsigned)p.L2 : (unsigned)img * (unsigned)p.L1;
        ldk_b = (unsigned)(bank ? p.ldk2 : p.ldk) * 2u, ldv_b = (unsigned)(bank ? p.ldvt2 : p.ldvt) * 2u;  // bytes
        kbase = reinterpret_cast<const char*>(bank ? p.K2 : p.K) + ((size_t)rowbase * ldk_b + (size_t)(head * D) * 2u);
        vbase = reinterpret_cast<const char*>(bank ? p.Vt2 : p.Vt) + ((size_t)(head * D) * ldv_b + (size_t)rowbase * 2u);
        kstep = 64u * ldk_b;
        src_kv0 = 0;
        src_L = bank ? p.L2 : p.L1;
        koff = hv_umul24((unsigned)kkey, ldk_b) + (unsigned)kpc * 16u;
        voff = hv_umul24((unsigned)vrow, ldv_b) + (unsigned)vkc * 16u;
    };
    auto request_tile = [&](int ti, int buf) {  // tiles are requested in order
        if (ti == T1) set_source(true);  // (rare, wave-uniform) the bank follows the own keys
        unsigned ko = koff, vo = voff;
        if (MASK && src_kv0 + 64 > src_L) {
            // ragged last tile: rows / key chunks beyond the end re-read the last valid one (in bounds; their scores are
            // masked to -inf and their probabilities are zero, so the values never count)
            const int last = src_L - 1 - src_kv0;  // last valid key of the tile (>= 0: L % 8 == 0, L > kv0)
            ko = hv_umul24((unsigned)min(kkey, last), ldk_b) + (unsigned)kpc * 16u;
            vo = hv_umul24((unsigned)vrow, ldv_b) + (unsigned)min(vkc, last >> 3) * 16u;
        }
        if (k_loader) hv_glds16_s(kbase, ko, Ks + buf * G::KSTRIDE + wave * 1024);
        if (v_loader) hv_glds16_s(vbase, vo, Vs + buf * G::VSTRIDE + (wave - 3) * 1024);
        kbase += kstep;
        vbase += 128;
        src_kv0 += 64;
    };
    // fragment read offsets (loop-invariant, relative to the buffer):
    //  K, A operand of the 32x32x16 MFMA: lane = key row l32 of the 32-key block, bytes 32 s + 16 half of its slice; the third
    //  step's upper half is the augmented column = the constant chunk behind the tile
    const int krd = l32 * G::KRS + half * 16;
    const int krd2_0 = half ? G::KBYTES : krd + 64, krd2_1 = half ? G::KBYTES : krd + 32 * G::KRS + 64;
    //  V^T, A operand of the 16x16x32 MFMA: lane = channel row 16 dt + r16, keys 32 kbk + 8 quad .. + 7
    int vrd[3][2];
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) {
            const int row = 16 * dt + r16;
            // rows 41 - 47 read a zeros row of their own bank half (odd: row 41's place, even: row 42's) at their own position
            vrd[dt][kbk] = (row <= D ? row : D + 2 - (row & 1)) * G::VRS + (((quad + 4 * kbk) ^ (row & 7)) << 4);
        }

    f32x4 oacc[2][3];  // O^T accumulators [16-query tile][16-row channel fragment]: lane = query r16, channels 16 dt + 4 quad + 0..3
    __syncthreads();  // LDS initialisation complete before the first tile store
    // pass 0: optimistic (the first tile's maximum is the reference throughout); pass 1, only after an overflow: careful
    for (int pass = 0; pass < 2; ++pass) {
    const bool careful = pass == 1;
    if (careful) {  // start over: the query operand's augmented element back to -m = 0
        mneg = 0.f;
        if (half) qf[2][0] = (short)0;
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) oacc[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    set_source(false);
    request_tile(0, 0);
    for (int ti = 0; ti < ntiles; ++ti) {
        const int buf = ti & 1;
        hv_vm_wait<0>();   // this wave's pieces of tile ti have landed (requested a whole tile ago) ...
        __syncthreads();   // ... and everybody's; every wave is done reading the other buffer
        if (ti + 1 < ntiles) request_tile(ti + 1, buf ^ 1);
        const unsigned char* kb = Ks + buf * G::KSTRIDE;
        const unsigned char* vb = Vs + buf * G::VSTRIDE;
        const int tile_kv0 = MASK ? ((ti >= T1 ? ti - T1 : ti) * 64) : 0;
        const int tile_L = MASK ? (ti >= T1 ? p.L2 : p.L1) : 0;
        auto mask_tail = [&](f32x16 (&sc)[2]) __attribute__((always_inline)) {
            if (!MASK) return;
            if (tile_kv0 + 64 > tile_L) {
#pragma unroll
                for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (tile_kv0 + 32 * kbk + 16 * half + i >= tile_L) sc[kbk][i] = -INFINITY;  // register i of half b = key 16 b + i
            }
        };
        auto scores = [&](f32x16 (&sc)[2]) __attribute__((always_inline)) {
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk) {
                f32x16 a;
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = 0.f;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int ofs = s < 2 ? krd + (32 * kbk) * G::KRS + s * 32 : (kbk ? krd2_1 : krd2_0);
                    const bf16x8 kf = hv_as_bf16x8(hv_ld16(kb + ofs));
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], a, 0, 0, 0);
                }
                sc[kbk] = a;
            }
        };
        // ---- S^T (two 32-key blocks x this wave's 32 queries), relative to the reference maximum
        f32x16 sacc[2];
        scores(sacc);
        mask_tail(sacc);
        // ---- probabilities (exp2 domain), packed to bf16 pairs: w[kbk][j] = keys 16 half + 2 j, + 1 of block kbk
        unsigned w[2][8];
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                w[kbk][j] = hv_pack2(__builtin_amdgcn_exp2f(sacc[kbk][2 * j]), __builtin_amdgcn_exp2f(sacc[kbk][2 * j + 1]));
        // careful pass only: some probability of the wave above 2^THR (bf16 256.0 = 0x4380; +inf = 0x7F80 is above as well)?
        // Positive bf16 order like 16-bit unsigned integers: the maximum runs on the packed pairs.
        bool over = false;
        if (careful) {  // (wave-uniform)
            unsigned pm = hv_pk_max_u16(w[0][0], w[0][1]);
#pragma unroll
            for (int j = 2; j < 8; ++j) pm = hv_pk_max_u16(pm, w[0][j]);
#pragma unroll
            for (int j = 0; j < 8; ++j) pm = hv_pk_max_u16(pm, w[1][j]);
            over = __any((pm & 0xffffu) > 0x4380u || (pm >> 16) > 0x4380u);
        }
        const bool first = ti == 0;  // the first tile fixes the reference maximum (it starts at 0, not at a score)
        if (first || over) {
            // rare: raise the reference maximum by the query's tile maximum, redo the exponentials against it and scale what
            // is still at the old reference (O^T with its denominator row) exactly once.  The scores are recomputed from LDS.
#ifndef HV_EMU
            asm volatile("" ::: "memory");  // keeps the LDS reads (and with them the MFMAs) below inside the branch
#endif
            f32x16 s2[2];
            scores(s2);
            mask_tail(s2);
            float mx = s2[0][0];
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s2[kbk][i]);
            mx = fmaxf(mx, hv_swap32(mx));  // the other half of the keys of the same query sits in lane ^ 32
            const float inc = first ? mx : fmaxf(mx, 0.f);
            const float mneg_new = hv_bf2f(hv_f2bf(mneg - inc));  // the new reference as the Q operand will carry it
            const float inc_eff = mneg - mneg_new;
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    w[kbk][j] = hv_pack2(__builtin_amdgcn_exp2f(s2[kbk][2 * j] - inc_eff), __builtin_amdgcn_exp2f(s2[kbk][2 * j + 1] - inc_eff));
            if (!first) {
                const float alpha = __builtin_amdgcn_exp2f(-inc_eff);
                const float a0 = __shfl(alpha, r16), a1 = __shfl(alpha, 16 + r16);  // lanes 0-31 hold queries 0-31 of the wave
#pragma unroll
                for (int dt = 0; dt < 3; ++dt) {
                    oacc[0][dt] *= a0;
                    oacc[1][dt] *= a1;
                }
            }
            mneg = mneg_new;
            if (half) qf[2][0] = (short)hv_f2bf(mneg_new);
        }
        // ---- P^T in the B-operand layout of the 16x16x32 MFMA: pf[qt][kbk], four lane-row swaps per 32-key block
        bf16x8 pf[2][2];
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) {
            u32x4 b0, b1;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                unsigned a = w[kbk][d], b = w[kbk][d + 4];
                hv_swap16_pair(a, b);
                b0[d] = a;
                b1[d] = b;
            }
            pf[0][kbk] = hv_as_bf16x8(b0);
            pf[1][kbk] = hv_as_bf16x8(b1);
        }
        // ---- O^T += V^T . P^T   (row 40 of V^T is all ones: accumulates the denominator)
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk) {
                const bf16x8 vf = hv_as_bf16x8(hv_ld16(vb + vrd[dt][kbk]));
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
                    oacc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qt][kbk], oacc[qt][dt], 0, 0, 0);
            }
    }
    if (careful) break;
    // ---- did anything overflow?  (inf / NaN anywhere in this lane's O^T columns, denominator row included: 0 x inf = NaN)
    float chk = 0.f;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) chk += (oacc[qt][dt][0] + oacc[qt][dt][1] + oacc[qt][dt][2] + oacc[qt][dt][3]) * 0.f;
    const int wave_bad = __any(chk != chk);
    __syncthreads();  // every wave is done with the tile buffers: the first bytes of the K buffer take the votes
    if (lane == 0) reinterpret_cast<int*>(smem)[wave] = wave_bad;
    __syncthreads();
    int any_bad = 0;
#pragma unroll
    for (int i = 0; i < G::NW; ++i) any_bad |= reinterpret_cast<const int*>(smem)[i];
    if (!any_bad) break;
    __syncthreads();  // the votes are read before pass 1 stores its first tile over them
    }

    // ---- normalise and store: lane owns query 16 qt + r16, channels 16 dt + 4 quad + 0..3; the denominator is O^T row 40
    //      (fragment 2, rows 8 .. 11 of it = quad 2, register 0)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const float l = __shfl(oacc[qt][2][0], 32 + r16);
        const float inv = 1.0f / l;
        const int q = q_wave + 16 * qt + r16;
        if (q >= p.Lq) continue;
        bf16_t* dst = p.O + ((long)img * p.Lq + q) * p.ldo + head * D;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            const int d = 16 * dt + 4 * quad;
            if (d < D) {
                const u32x2 o = {hv_pack2(oacc[qt][dt][0] * inv, oacc[qt][dt][1] * inv),
                                 hv_pack2(oacc[qt][dt][2] * inv, oacc[qt][dt][3] * inv)};
                hv_st8(dst + d, o);
            }
        }
    }
}

// tuning knob (hv_set_tuning): 1 = this kernel for head dim 40 (default), 0 = the generic kernel
static int g_hv_attn40 = 1;

static inline void hv_attention40_launch(const hv_attention_params& p, hipStream_t stream) {
    using G = HvAttn40Geom;
    const int total = ((p.Lq + G::BQ - 1) / G::BQ) * p.heads * p.n_images;
    const int grid = ((total + 7) / 8) * 8;
    const bool ragged = (p.L1 % 64) != 0 || (p.L2 % 64) != 0;
    hv_note("hv_attention40_kernel | n=%d heads=%d D=%d Lq=%d L1=%d L2=%d bank=%d", p.n_images, p.heads, p.D, p.Lq, p.L1, p.L2,
            p.bank_sel != nullptr && p.L2 > 0);
    if (ragged) hv_launch(hv_attention40_kernel<true>, dim3(grid), dim3(512), stream, p);
    else hv_launch(hv_attention40_kernel<false>, dim3(grid), dim3(512), stream, p);
}
