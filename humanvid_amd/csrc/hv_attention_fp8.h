// hv_attention_fp8.h -- the spatial attention of hv_attention.h with both matrix products on the fp8 MFMA
// (v_mfma_f32_16x16x32_fp8_fp8, OCP e4m3 on gfx950) -- BASELINE.json configs[4] ("fp8 MFMA attention").
//
// Reference semantics are those of hv_attention.h (read-mode reference attention,
// /root/reference/src/models/mutual_self_attention.py:147-186: own keys || bank keys for the conditional images, own keys only
// for the CFG-unconditional ones); only the operand precision of QK^T and PV changes.  Quantisation, chosen so that no
// scale has to be applied inside an accumulation chain:
//   * K: one scale per (image | bank batch, head, 64-key tile) (amax / 384).
//   * Q: one scale per query row (amax over the head's channels / 384), computed in registers at load; the queries stay
//     resident as e4m3 fragments.  score (exp2 domain) = acc * (k_scale[tile] * q_scale[query] * scale * log2 e): one
//     fused multiply-subtract per score, the factor is lane-local because a lane owns one query column.
//   * P: probabilities are <= 1 after the running maximum is subtracted; they enter the PV product as e4m3(128 p).
//   * V: one scale per head (amax over all images and the bank) -- the O^T accumulation runs over all tiles of both key
//     sources without rescaling; the row of ones that yields the denominator is e4m3 1.0, so
//     O / l = (sum v8 p8) * v_scale / (sum p8) and the 128 cancels.
//   K and V are quantised ONCE per attention call by the pre-pass (hv_attention_fp8_quantize: amax launch + quantise launch per
//   key source) into e4m3 copies; the attention kernel streams those (half the bytes of the bf16 tiles, no conversion in its
//   tile loop -- every K / V tile is re-read by all 128-query workgroups of its image, 48 of them at level 0).
// Accuracy: e4m3 carries 3 mantissa bits (2^-4 relative); errors average over the head dim in QK^T and over the keys in PV.
// Stated and tested bound (tests/kernel_cases.py::case_attention_fp8): NRMSE <= 3e-2 against fp32 SDPA on bf16-rounded inputs
// (the bf16 kernel: <= 6e-3).
//
// Structure = hv_attention_kernel (S^T = K.Q^T, lane owns a query column, key rows permuted in LDS so that P^T fragments feed
// V^T.P^T without cross-lane moves, register-staged double-buffered K / V^T tiles, one barrier per tile, online softmax,
// lazy rescale); fragments are 8 bytes per lane instead of 16, so a tile costs half the LDS bytes.
#pragma once
#include "hv_common.h"
#include "humanvid_hip.h"

#define HV_FP8_AMAX_TARGET 384.0f  // e4m3 maximum is 448: keep headroom for the rounding of the scale itself
#define HV_FP8_P_SCALE 128.0f

#ifndef HV_EMU
typedef long hv_fp8x8;  // 8 e4m3 values, the A / B operand of v_mfma_f32_16x16x32_fp8_fp8
HV_DEV unsigned hv_cvt_pk_fp8(float a, float b, unsigned old, bool hi) {  // hi: a compile-time constant after inlining
    return hi ? (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, true)
              : (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, false);
}
HV_DEV f32x4 hv_mfma_fp8(hv_fp8x8 a, hv_fp8x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, c, 0, 0, 0); }
#else
typedef long hv_fp8x8;
HV_DEV unsigned hv_cvt_pk_fp8(float a, float b, unsigned old, bool hi) { return hvemu::cvt_pk_fp8(a, b, old, hi); }
HV_DEV f32x4 hv_mfma_fp8(hv_fp8x8 a, hv_fp8x8 b, f32x4 c) { return hvemu::mfma_fp8(a, b, c); }
#endif

HV_DEV hv_fp8x8 hv_pack_fp8x8(const float* f) {  // 8 floats -> 8 e4m3 (element e in byte e)
    unsigned lo = 0, hi = 0;
    lo = hv_cvt_pk_fp8(f[0], f[1], lo, false);
    lo = hv_cvt_pk_fp8(f[2], f[3], lo, true);
    hi = hv_cvt_pk_fp8(f[4], f[5], hi, false);
    hi = hv_cvt_pk_fp8(f[6], f[7], hi, true);
    return (hv_fp8x8)(((unsigned long)hi << 32) | lo);
}

// ---- pre-pass, two launches over one key source (the keys / values of an image are re-read by every 128-query workgroup of
//      the attention kernel -- 48 times at level 0 -- so they are quantised ONCE, here, not in the attention kernel's tile loop):
//   (1) amax: K scale per (image, head, 64-key tile), V amax per head (integer atomic max over images and tiles of the
//       non-negative float bit patterns; the launcher zero-fills vamax[heads]);
//   (2) quantise: K8[(img*L + kv)*ldk8 + h*D + d] = e4m3(K / kscale[tile]),  Vt8[(h*D + d)*ldvt8 + img*L + kv] = e4m3(V / vscale)
//       with vscale = max(vamax[h], vfloor[h]) / 384 -- vfloor = the OTHER key source's amax (own values <-> bank), so that both
//       sources of a launch carry one V scale per head and the O^T accumulation never changes scale.
// One 64-thread workgroup per (tile, head, image); thread = key row for K, (8-token chunk, channel stripe) for V^T.
template <int D>
__global__ __launch_bounds__(64) void hv_attention_fp8_amax_kernel(const bf16_t* K, long ldk, const bf16_t* Vt, long ldvt, int L,
                                                                   int heads, float* kscale, float* vamax) {
    const int tile = blockIdx.x, head = blockIdx.y, img = blockIdx.z;
    const int T = (L + 63) / 64;
    const int t = threadIdx.x;
    const int kv = tile * 64 + t;
    float ka = 0.f, va = 0.f;
    if (kv < L) {
        const bf16_t* krow = K + ((long)img * L + kv) * ldk + head * D;
#pragma unroll
        for (int c = 0; c < D / 8; ++c) {
            float f[8];
            hv_unpack8(hv_ld16(krow + c * 8), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) ka = fmaxf(ka, fabsf(f[e]));
        }
    }
    const int c8 = t & 7;
    if (tile * 64 + c8 * 8 < L) {
        for (int d = t >> 3; d < D; d += 8) {
            float f[8];
            hv_unpack8(hv_ld16(Vt + (long)(head * D + d) * ldvt + (long)img * L + tile * 64 + c8 * 8), f);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (tile * 64 + c8 * 8 + e < L) va = fmaxf(va, fabsf(f[e]));
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        ka = fmaxf(ka, __shfl_xor(ka, m));
        va = fmaxf(va, __shfl_xor(va, m));
    }
    if (t == 0) {
        kscale[((long)img * heads + head) * T + tile] = ka > 0.f ? ka / HV_FP8_AMAX_TARGET : 1.0f;
        atomicMax(reinterpret_cast<int*>(vamax) + head, __builtin_bit_cast(int, va));  // one V scale per head
    }
}

HV_DEV float hv_fp8_vscale(float amax) { return amax > 0.f ? amax / HV_FP8_AMAX_TARGET : 1.0f; }

template <int D>
__global__ __launch_bounds__(64) void hv_attention_fp8_quant_kernel(const bf16_t* K, long ldk, const bf16_t* Vt, long ldvt, int L,
                                                                    int heads, const float* kscale, const float* vamax,
                                                                    const float* vfloor, unsigned char* K8, long ldk8,
                                                                    unsigned char* Vt8, long ldvt8) {
    const int tile = blockIdx.x, head = blockIdx.y, img = blockIdx.z;
    const int T = (L + 63) / 64;
    const int t = threadIdx.x;
    const int kv = tile * 64 + t;
    const float inv_k = 1.0f / kscale[((long)img * heads + head) * T + tile];
    float vam = vamax[head];
    if (vfloor != nullptr) vam = fmaxf(vam, vfloor[head]);
    const float inv_v = 1.0f / hv_fp8_vscale(vam);
    if (kv < L) {
        const bf16_t* krow = K + ((long)img * L + kv) * ldk + head * D;
        unsigned char* orow = K8 + ((long)img * L + kv) * ldk8 + head * D;
#pragma unroll
        for (int c = 0; c < D / 8; ++c) {
            float f[8];
            hv_unpack8(hv_ld16(krow + c * 8), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] *= inv_k;
            const hv_fp8x8 o = hv_pack_fp8x8(f);
            hv_st8(orow + c * 8, u32x2{(unsigned)o, (unsigned)((unsigned long)o >> 32)});
        }
    }
    const int c8 = t & 7;
    if (tile * 64 + c8 * 8 < L) {
        for (int d = t >> 3; d < D; d += 8) {
            const long col = (long)img * L + tile * 64 + c8 * 8;
            float f[8];
            hv_unpack8(hv_ld16(Vt + (long)(head * D + d) * ldvt + col), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] *= inv_v;
            const hv_fp8x8 o = hv_pack_fp8x8(f);
            hv_st8(Vt8 + (long)(head * D + d) * ldvt8 + col, u32x2{(unsigned)o, (unsigned)((unsigned long)o >> 32)});
        }
    }
}

template <int D>
struct HvAttn8Geom {
    static constexpr int NFULL = (D + 31) / 32;       // 32-deep QK^T steps (zero-padded remainder)
    static constexpr int DT = (D + 15) / 16;          // 16-row fragments of V^T / O^T
    static constexpr int DK = 32 * NFULL;             // bytes of a key row in LDS
    static constexpr int DV = 16 * DT;
    static constexpr bool ONES = DV > D;              // spare V^T row for the denominator
    static constexpr int KRS = DK + 8;                // K row stride (bytes): 8-byte reads of 16 rows hit 32 distinct banks
    static constexpr int VRS = 64 + 8;                // V^T row stride (64 keys)
    static constexpr int KBYTES = 64 * KRS;
    static constexpr int VBYTES = DV * VRS;
    static constexpr int KCH = 64 * (D / 8);          // 8-byte e4m3 chunks of a K tile
    static constexpr int VCH = D * 8;
    static constexpr int KIT = (KCH + 255) / 256;
    static constexpr int VIT = (VCH + 255) / 256;
    static constexpr int QT = 2;
    static constexpr int BQ = 4 * 16 * QT;
};

template <int D, bool MASK>
__global__ __launch_bounds__(256, (D == 40 ? 4 : (D == 80 ? 2 : 1))) void hv_attention_fp8_kernel(
    hv_attention_params p, const float* kscale, const float* vamax, const float* kscale2, const float* vamax2) {
    using G = HvAttn8Geom<D>;
    constexpr int NFULL = G::NFULL, DT = G::DT, QT = G::QT;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (G::KBYTES + G::VBYTES)];
    unsigned char* Ks = smem;
    unsigned char* Vs = smem + 2 * G::KBYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, quad = lane >> 4;

    const int nqb = (p.Lq + G::BQ - 1) / G::BQ;
    const int total = nqb * p.heads * p.n_images;
    const int cpx = gridDim.x / 8;
    int t = (blockIdx.x % 8) * cpx + blockIdx.x / 8;
    if (t >= total) return;
    const int qb = t % nqb;
    t /= nqb;
    const int head = t % p.heads;
    int img = t / p.heads;
    if ((p.n_images & 1) == 0) img = (img & 1) * (p.n_images >> 1) + (img >> 1);  // CFG halves alternate over the XCDs
    const int sel = (p.bank_sel != nullptr && p.L2 > 0) ? p.bank_sel[img] : -1;
    const int T1 = (p.L1 + 63) / 64;
    const int T2 = sel >= 0 ? (p.L2 + 63) / 64 : 0;
    const int ntiles = T1 + T2;

    for (int i = tid; i < 2 * (G::KBYTES + G::VBYTES) / 16; i += 256) hv_st16(smem + i * 16, u32x4{0u, 0u, 0u, 0u});
    __syncthreads();
    if (G::ONES) {  // e4m3 1.0 = 0x38 in the first spare V^T row of both buffers
        for (int i = tid; i < 2 * 8; i += 256)
            hv_st8(Vs + (i >> 3) * G::VBYTES + D * G::VRS + (i & 7) * 8, u32x2{0x38383838u, 0x38383838u});
    }

    // ---- queries: per-row scale, e4m3 fragments resident for the whole kernel
    hv_fp8x8 qf[QT][NFULL];
    float qfac[QT];  // q_scale * softmax scale * log2(e)
    const int q_wave = qb * G::BQ + wave * 16 * QT;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int q = q_wave + 16 * qt + r16;
        const bf16_t* qrow = p.Q + ((long)img * p.Lq + q) * p.ldq + head * D;
        float f[NFULL][8];
        float am = 0.f;
#pragma unroll
        for (int s = 0; s < NFULL; ++s) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (q < p.Lq && 32 * s + 8 * quad + 8 <= D) v = hv_ld16(qrow + 32 * s + 8 * quad);
            hv_unpack8(v, f[s]);
#pragma unroll
            for (int e = 0; e < 8; ++e) am = fmaxf(am, fabsf(f[s][e]));
        }
        am = fmaxf(am, __shfl_xor(am, 16));
        am = fmaxf(am, __shfl_xor(am, 32));
        const float qs = am > 0.f ? am / HV_FP8_AMAX_TARGET : 1.0f;
        const float inv = 1.0f / qs;
#pragma unroll
        for (int s = 0; s < NFULL; ++s) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[s][e] *= inv;
            qf[qt][s] = hv_pack_fp8x8(f[s]);
        }
        qfac[qt] = qs * p.scale * 1.44269504089f;
    }

    // one V scale per head for the whole launch: own values and bank values were quantised with
    // max(amax_own[h], amax_bank[h]) / 384 (each quantiser call gets the other source's amax as its floor)
    float vam = vamax[head];
    if (p.bank_sel != nullptr && p.L2 > 0) vam = fmaxf(vam, vamax2[head]);
    const float vs = hv_fp8_vscale(vam);

    const unsigned char* K8 = reinterpret_cast<const unsigned char*>(p.K);
    const unsigned char* V8 = reinterpret_cast<const unsigned char*>(p.Vt);
    const unsigned char* K82 = reinterpret_cast<const unsigned char*>(p.K2);
    const unsigned char* V82 = reinterpret_cast<const unsigned char*>(p.Vt2);
    u32x2 kreg[G::KIT], vreg[G::VIT];
    float ksc_staged = 1.f;  // scale of the K tile held in kreg
    auto load_tile = [&](int ti) {
        const bool bank = ti >= T1;
        const int tl = bank ? ti - T1 : ti;
        const int kv0 = tl * 64;
        const int L = bank ? p.L2 : p.L1;
        const long rowbase = bank ? (long)sel * p.L2 : (long)img * p.L1;
        const unsigned char* Kp = bank ? K82 : K8;
        const long ldk = bank ? p.ldk2 : p.ldk;
        const unsigned char* Vp = bank ? V82 : V8;
        const long ldv = bank ? p.ldvt2 : p.ldvt;
        ksc_staged = bank ? kscale2[((long)sel * p.heads + head) * T2 + tl] : kscale[((long)img * p.heads + head) * T1 + tl];
#pragma unroll
        for (int i = 0; i < G::KIT; ++i) {
            const int id = tid + 256 * i;
            u32x2 v = {0u, 0u};
            if (id < G::KCH) {
                const int r = id / (D / 8), c = id % (D / 8);
                if (!MASK || kv0 + r < L) v = hv_ld8(Kp + (rowbase + kv0 + r) * ldk + head * D + c * 8);
            }
            kreg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < G::VIT; ++i) {
            const int id = tid + 256 * i;
            u32x2 v = {0u, 0u};
            if (id < G::VCH) {
                const int d = id >> 3, c = id & 7;
                if (!MASK || kv0 + c * 8 < L) v = hv_ld8(Vp + (long)(head * D + d) * ldv + rowbase + kv0 + c * 8);
            }
            vreg[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < G::KIT; ++i) {
            const int id = tid + 256 * i;
            if (id < G::KCH) {
                const int r = id / (D / 8), c = id % (D / 8);
                const int lr = (r & ~0x1c) | ((r & 4) << 2) | ((r & 0x18) >> 1);  // key permutation of hv_attention.h
                hv_st8(Ks + buf * G::KBYTES + lr * G::KRS + c * 8, kreg[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < G::VIT; ++i) {
            const int id = tid + 256 * i;
            if (id < G::VCH) hv_st8(Vs + buf * G::VBYTES + (id >> 3) * G::VRS + (id & 7) * 8, vreg[i]);
        }
    };

    f32x4 oacc[QT][DT];
    float mrun[QT], lrun[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        mrun[qt] = -INFINITY;
        lrun[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) oacc[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    load_tile(0);
    __syncthreads();
    for (int ti = 0; ti < ntiles; ++ti) {
        const int buf = ti & 1;
        const float sk = ksc_staged;  // scale of the K tile being parked now (load_tile below replaces it with the next one)
        store_tile(buf);
        __syncthreads();
        if (ti + 1 < ntiles) load_tile(ti + 1);
        const unsigned char* kb = Ks + buf * G::KBYTES + r16 * G::KRS + quad * 8;
        const unsigned char* vb = Vs + buf * G::VBYTES + r16 * G::VRS + quad * 8;

        // ---- S^T fragments (unscaled e4m3 products): sacc[kvf][qt], reg r <-> key 32*(kvf>>1) + 8*quad + 4*(kvf&1) + r
        f32x4 sacc[4][QT];
#pragma unroll
        for (int kvf = 0; kvf < 4; ++kvf) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) sacc[kvf][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NFULL; ++s) {
                const u32x2 kw = hv_ld8(kb + (16 * kvf) * G::KRS + s * 32);
                const hv_fp8x8 kf = (hv_fp8x8)(((unsigned long)kw[1] << 32) | kw[0]);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) sacc[kvf][qt] = hv_mfma_fp8(kf, qf[qt][s], sacc[kvf][qt]);
            }
        }
        if (MASK) {
            const bool bank = ti >= T1;
            const int kv0 = (bank ? ti - T1 : ti) * 64;
            const int L = bank ? p.L2 : p.L1;
            if (kv0 + 64 > L) {
#pragma unroll
                for (int kvf = 0; kvf < 4; ++kvf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kv = kv0 + 32 * (kvf >> 1) + 8 * quad + 4 * (kvf & 1) + r;
                        if (kv >= L) {
#pragma unroll
                            for (int qt = 0; qt < QT; ++qt) sacc[kvf][qt][r] = -INFINITY;
                        }
                    }
            }
        }
        // ---- online softmax (exp2 domain), probabilities as e4m3(128 p)
        hv_fp8x8 pf[QT][2];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const float fac = sk * qfac[qt];  // > 0: the maximum commutes with the scaling
            float mx = fmaxf(fmaxf(sacc[0][qt][0], sacc[0][qt][1]), fmaxf(sacc[0][qt][2], sacc[0][qt][3]));
#pragma unroll
            for (int kvf = 1; kvf < 4; ++kvf)
                mx = fmaxf(fmaxf(fmaxf(mx, sacc[kvf][qt][0]), fmaxf(sacc[kvf][qt][1], sacc[kvf][qt][2])), sacc[kvf][qt][3]);
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mold = mrun[qt];
            const float mnew = fmaxf(mold, mx * fac);
            mrun[qt] = mnew;
            float pv[4][4];
            float psum = 0.f;
#pragma unroll
            for (int kvf = 0; kvf < 4; ++kvf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // (+7: the factor 128 of the e4m3 probabilities, inside the exponent)
                    pv[kvf][r] = __builtin_amdgcn_exp2f(sacc[kvf][qt][r] * fac - mnew + 7.0f);
                    if (!G::ONES) psum += pv[kvf][r];
                }
            if (__any(mnew > mold)) {
                const float alpha = __builtin_amdgcn_exp2f(mold - mnew);
                if (!G::ONES) lrun[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) oacc[qt][dt] *= alpha;
            }
            if (!G::ONES) lrun[qt] += psum;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const float f8[8] = {pv[2 * ks][0],     pv[2 * ks][1],     pv[2 * ks][2],     pv[2 * ks][3],
                                     pv[2 * ks + 1][0], pv[2 * ks + 1][1], pv[2 * ks + 1][2], pv[2 * ks + 1][3]};
                pf[qt][ks] = hv_pack_fp8x8(f8);
            }
        }
        // ---- O^T += V^T . P^T  (row D of V^T is e4m3 1.0 when ONES: accumulates the denominator)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const u32x2 vw = hv_ld8(vb + (16 * dt) * G::VRS + ks * 32);
                const hv_fp8x8 vf = (hv_fp8x8)(((unsigned long)vw[1] << 32) | vw[0]);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) oacc[qt][dt] = hv_mfma_fp8(vf, pf[qt][ks], oacc[qt][dt]);
            }
    }

    // ---- normalise and store: lane owns query r16, channels 16*dt + 4*quad + 0..3
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l;
        if (G::ONES) {
            l = __shfl(oacc[qt][D / 16][D % 4], ((D % 16) / 4) * 16 + r16);
        } else {
            // the probabilities were rounded to e4m3 before the PV product: sum the rounded values' exact counterparts is not
            // available without the ones row; d = 80 / 160 have no spare row, their denominator is the fp32 sum (scaled 128)
            l = lrun[qt];
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
        }
        const float inv = vs / l;
        const int q = q_wave + 16 * qt + r16;
        if (q >= p.Lq) continue;
        bf16_t* dst = p.O + ((long)img * p.Lq + q) * p.ldo + head * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = 16 * dt + 4 * quad;
            if (d < D) {
                u32x2 o = {hv_pack2(oacc[qt][dt][0] * inv, oacc[qt][dt][1] * inv),
                           hv_pack2(oacc[qt][dt][2] * inv, oacc[qt][dt][3] * inv)};
                hv_st8(dst + d, o);
            }
        }
    }
}

// zeroes the per-head V amax ahead of the atomicMax pass.  A kernel (not hipMemsetAsync) so that it goes through hv_launch:
// the command-list recorder of the multi-GPU step only sees hv_launch, and a memset it does not replay left the running
// maximum of every earlier layer and step in the shared buffer.
__global__ void hv_attention_fp8_zero_kernel(float* v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = 0.f;
}

template <int D>
static inline void hv_attention_fp8_quantize_launch_t(const bf16_t* K, long ldk, const bf16_t* Vt, long ldvt, int n, int heads, int L,
                                                      float* kscale, float* vamax, const float* vfloor, unsigned char* K8,
                                                      long ldk8, unsigned char* Vt8, long ldvt8, int phase, hipStream_t stream) {
    const dim3 grid((L + 63) / 64, heads, n);
    if (phase & 1) {
        hv_note("hv_attention_fp8_amax_kernel<%d> | n=%d heads=%d L=%d", D, n, heads, L);
        hv_launch(hv_attention_fp8_amax_kernel<D>, grid, dim3(64), stream, K, ldk, Vt, ldvt, L, heads, kscale, vamax);
    }
    if (phase & 2) {
        hv_note("hv_attention_fp8_quant_kernel<%d> | n=%d heads=%d L=%d", D, n, heads, L);
        hv_launch(hv_attention_fp8_quant_kernel<D>, grid, dim3(64), stream, K, ldk, Vt, ldvt, L, heads, (const float*)kscale,
                  (const float*)vamax, vfloor, K8, ldk8, Vt8, ldvt8);
    }
}

// phase: 1 = amax only, 2 = quantise only (kscale / vamax from an earlier amax call), 3 = both
static inline int hv_attention_fp8_quantize_launch(const bf16_t* K, long ldk, const bf16_t* Vt, long ldvt, int n, int heads, int D,
                                                   int L, float* kscale, float* vamax, const float* vfloor, unsigned char* K8,
                                                   long ldk8, unsigned char* Vt8, long ldvt8, int phase, hipStream_t stream) {
    if (L <= 0 || L % 8 != 0 || ldk % 8 || ldvt % 8 || n <= 0 || heads <= 0 || phase < 1 || phase > 3) return -1;
    if ((phase & 2) && (!K8 || !Vt8 || ldk8 % 8 || ldvt8 % 8)) return -1;
    if (phase & 1) {
        hv_note("hv_attention_fp8_zero_kernel | heads=%d", heads);
        hv_launch(hv_attention_fp8_zero_kernel, dim3((heads + 63) / 64), dim3(64), stream, vamax, heads);
    }
    switch (D) {
        case 40: hv_attention_fp8_quantize_launch_t<40>(K, ldk, Vt, ldvt, n, heads, L, kscale, vamax, vfloor, K8, ldk8, Vt8, ldvt8, phase, stream); break;
        case 80: hv_attention_fp8_quantize_launch_t<80>(K, ldk, Vt, ldvt, n, heads, L, kscale, vamax, vfloor, K8, ldk8, Vt8, ldvt8, phase, stream); break;
        case 160: hv_attention_fp8_quantize_launch_t<160>(K, ldk, Vt, ldvt, n, heads, L, kscale, vamax, vfloor, K8, ldk8, Vt8, ldvt8, phase, stream); break;
        default: return -2;
    }
    return 0;
}

template <int D>
static inline void hv_attention_fp8_launch_t(const hv_attention_params& p, const float* ks, const float* va, const float* ks2,
                                             const float* va2, hipStream_t stream) {
    using G = HvAttn8Geom<D>;
    const int total = ((p.Lq + G::BQ - 1) / G::BQ) * p.heads * p.n_images;
    const int grid = ((total + 7) / 8) * 8;
    const bool ragged = (p.L1 % 64) != 0 || (p.L2 % 64) != 0;
    hv_note("hv_attention_fp8_kernel<%d> | n=%d heads=%d D=%d Lq=%d L1=%d L2=%d bank=%d", D, p.n_images, p.heads, D, p.Lq, p.L1, p.L2,
            p.bank_sel != nullptr && p.L2 > 0);
    if (ragged)
        hv_launch(hv_attention_fp8_kernel<D, true>, dim3(grid), dim3(256), stream, p, ks, va, ks2, va2);
    else
        hv_launch(hv_attention_fp8_kernel<D, false>, dim3(grid), dim3(256), stream, p, ks, va, ks2, va2);
}

static inline int hv_attention_fp8_launch(const hv_attention_params& p, const float* ks, const float* va, const float* ks2,
                                          const float* va2, hipStream_t stream) {
    if (p.L1 <= 0 || p.L1 % 8 != 0 || p.L2 % 8 != 0 || p.Lq <= 0) return -1;
    if (p.ldq % 8 || p.ldk % 8 || p.ldvt % 8 || p.ldo % 4) return -1;
    const bool bank = p.L2 > 0 && p.bank_sel != nullptr;
    if (bank && (!p.K2 || !p.Vt2 || p.ldk2 % 8 || p.ldvt2 % 8 || !ks2 || !va2)) return -1;
    switch (p.D) {
        case 40: hv_attention_fp8_launch_t<40>(p, ks, va, ks2, va2, stream); break;
        case 80: hv_attention_fp8_launch_t<80>(p, ks, va, ks2, va2, stream); break;
        case 160: hv_attention_fp8_launch_t<160>(p, ks, va, ks2, va2, stream); break;
        default: return -2;
    }
    return 0;
}
