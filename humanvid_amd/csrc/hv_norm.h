// hv_norm.h -- normalisation statistics (HBM-bound reductions); the normalise/affine step itself
// is always fused into the consumer kernel (hv_conv3x3 / hv_gemm), never materialised.
//
//  * GroupNorm(32) per image (InflatedGroupNorm, /root/reference/src/models/resnet.py:18-26;
//    torch.nn.GroupNorm in transformer_3d.py:58-60 and motion_module.py:119-121): one pass over
//    the channels-last activation with 16-byte loads, pivoted fp32 sums per channel in registers,
//    merged to groups / images as (mean, M2) pairs (Chan / Welford); deterministic (no atomics); a tiny
//    second kernel turns (mean, rstd, gamma, beta) into per-(image,channel) scale/shift.
//    Supports the two-source channel concat of the up-blocks (groups may straddle the seam).
//  * LayerNorm row statistics (mean, rstd), one wavefront per token row, two-pass in registers.
#pragma once
#include "hv_common.h"
#include "humanvid_hip.h"

#define HV_GN_MAXREP 2  // channel-vectors per thread: supports C <= 256 * 8 * 2 = 4096

// Numerics: variance from raw fp32 sums (E[x^2] - E[x]^2) loses all its digits once |mean| >> std, which real
// activations do reach in single channels.  Here every channel is accumulated around a pivot K_c (its own value at pixel
// 0 of the image, identical in every workgroup and lane), i.e. sum(x - K_c) and sum((x - K_c)^2): K_c lies inside the
// channel's range, so M2_c = q - s^2/n cancels at most a few std^2.  Channels -> group and pixel ranges -> image are then
// merged as (count, mean, M2) triples with Chan's pairwise update -- the parallel form of Welford's algorithm
// (SURVEY.md 7): no step ever subtracts two quantities of the size of mean^2.
__global__ __launch_bounds__(256) void hv_gn_partial_kernel(hv_groupnorm_params p) {
    // grid: (splits, n_images).  Each workgroup reduces a contiguous pixel range of one image.
    __shared__ float red[3][4096];  // per channel: sum(x - K), sum((x - K)^2), K
    const int tid = threadIdx.x;
    const int C = p.C1 + p.C2, CV = C / 8;
    const int img = blockIdx.y, split = blockIdx.x;
    const int per = (p.pixels + p.splits - 1) / p.splits;
    const int pb = split * per, pe = min(pb + per, p.pixels);

    // thread -> (channel vector, pixel lane)
    const int P = CV <= 256 ? 256 / CV : 1;   // pixels processed concurrently
    const int nrep = CV <= 256 ? 1 : (CV + 255) / 256;
    float s[HV_GN_MAXREP][8], q[HV_GN_MAXREP][8], piv[HV_GN_MAXREP][8];
#pragma unroll
    for (int r = 0; r < HV_GN_MAXREP; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) s[r][e] = q[r][e] = piv[r][e] = 0.f;

    const int cv0 = CV <= 256 ? tid % CV : tid;
    const int pl = CV <= 256 ? tid / CV : 0;
    const bool active = pl < P;
    auto src_of = [&](int pix, int c) {
        return c < p.C1 ? p.X + ((long)img * p.pixels + pix) * p.C1 + c
                        : p.X2 + ((long)img * p.pixels + pix) * p.C2 + (c - p.C1);
    };
    if (active) {
#pragma unroll
        for (int r = 0; r < HV_GN_MAXREP; ++r) {
            const int cv = cv0 + 256 * r;
            if (r < nrep && cv < CV) hv_unpack8(hv_ld16(src_of(0, cv * 8)), piv[r]);
        }
        int pix = pb + pl;
        // four pixels per trip, all loads issued before the first is consumed (one 16-byte load in flight per thread left the
        // pass at 2.2-3.1 TB/s: latency-bound); single-vector channels only (C <= 2048), the tail and wide C below
        if (nrep == 1 && cv0 < CV) {
            for (; pix + 3 * P < pe; pix += 4 * P) {
                u32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = hv_ld16(src_of(pix + u * P, cv0 * 8));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float f[8];
                    hv_unpack8(v[u], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d = f[e] - piv[0][e];
                        s[0][e] += d;
                        q[0][e] += d * d;
                    }
                }
            }
        }
        for (; pix < pe; pix += P) {
#pragma unroll
            for (int r = 0; r < HV_GN_MAXREP; ++r) {
                const int cv = cv0 + 256 * r;
                if (r < nrep && cv < CV) {
                    float f[8];
                    hv_unpack8(hv_ld16(src_of(pix, cv * 8)), f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d = f[e] - piv[r][e];
                        s[r][e] += d;
                        q[r][e] += d * d;
                    }
                }
            }
        }
    }
    // reduce across pixel lanes: lane pl == 0 initialises, the others add in turn (P <= 64 rounds
    // only when C is tiny; typical P is 1..6)
    for (int round = 0; round < P; ++round) {
        if (active && pl == round) {
#pragma unroll
            for (int r = 0; r < HV_GN_MAXREP; ++r) {
                const int cv = cv0 + 256 * r;
                if (r < nrep && cv < CV) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (round == 0) {
                            red[0][cv * 8 + e] = s[r][e];
                            red[1][cv * 8 + e] = q[r][e];
                            red[2][cv * 8 + e] = piv[r][e];
                        } else {
                            red[0][cv * 8 + e] += s[r][e];
                            red[1][cv * 8 + e] += q[r][e];
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    // channels of a group (equal counts n) -> (mean, M2) of this pixel range
    const int cg = C / p.groups;
    const float n = (float)max(pe - pb, 0);
    for (int g = tid; g < p.groups; g += 256) {
        float mean_g = 0.f, m2 = 0.f;
        if (n > 0.f) {
            float msum = 0.f;
            for (int c = g * cg; c < (g + 1) * cg; ++c) msum += red[2][c] + red[0][c] / n;
            mean_g = msum / (float)cg;
            for (int c = g * cg; c < (g + 1) * cg; ++c) {
                const float sc = red[0][c];
                const float dm = red[2][c] + sc / n - mean_g;
                m2 += (red[1][c] - sc * sc / n) + n * dm * dm;
            }
        }
        float* dst = p.partial + (((long)img * p.splits + split) * p.groups + g) * 2;
        dst[0] = mean_g;
        dst[1] = m2;
    }
}

__global__ __launch_bounds__(256) void hv_gn_finalize_kernel(hv_groupnorm_params p) {
    // grid: (ceil(groups / 4), n_images): one wavefront per (image, group), lane = pixel range (splits <= 64), so that the
    // merge is one load per lane and two butterfly sums instead of two serial loops over the ranges (round 2: 82 launches
    // per step, 14-23 us each as a per-channel serial merge)
    const int C = p.C1 + p.C2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.x * 4 + wave, img = blockIdx.y;
    const int cg = C / p.groups;
    const bool live = g < p.groups;  // no early return: every lane takes part in the wave shuffles below
    const int per = (p.pixels + p.splits - 1) / p.splits;
    // merge the pixel ranges: mean = sum n_i mean_i / N,  M2 = sum M2_i + sum n_i (mean_i - mean)^2
    float ni = 0.f, mi = 0.f, qi = 0.f;
    if (live && lane < p.splits) {
        ni = (float)max(min(per, p.pixels - lane * per), 0);
        const float* src = p.partial + (((long)img * p.splits + lane) * p.groups + g) * 2;
        mi = src[0];
        qi = src[1];
    }
    float wsum = ni * mi;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wsum += __shfl_xor(wsum, o);
    const float mean = wsum / (float)p.pixels;
    const float dm = mi - mean;
    float m2 = qi + ni * (float)cg * dm * dm;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m2 += __shfl_xor(m2, o);
    const float var = m2 / ((float)cg * (float)p.pixels);
    const float rstd = 1.0f / sqrtf(var + p.eps);
    if (!live) return;
    for (int c = g * cg + lane; c < (g + 1) * cg; c += 64) {
        const float sc = rstd * p.gamma[c];
        p.scale[(long)img * C + c] = sc;
        p.shift[(long)img * C + c] = p.beta[c] - mean * sc;
    }
}

static inline int hv_groupnorm_launch(const hv_groupnorm_params& p, hipStream_t stream) {
    const int C = p.C1 + p.C2;
    if (p.C1 % 8 != 0 || p.C2 % 8 != 0 || C % p.groups != 0 || C > 4096 || p.splits < 1 || p.splits > 64) return -1;
    if (p.C2 > 0 && p.X2 == nullptr) return -1;
    hv_note("hv_gn_partial_kernel | n=%d pixels=%d C=%d", p.n_images, p.pixels, C);
    hv_launch(hv_gn_partial_kernel, dim3(p.splits, p.n_images), dim3(256), stream, p);
    hv_note("hv_gn_finalize_kernel | n=%d C=%d", p.n_images, C);
    hv_launch(hv_gn_finalize_kernel, dim3((p.groups + 3) / 4, p.n_images), dim3(256), stream, p);
    return 0;
}

// ---- GroupNorm scale / shift from the partial statistics the producing kernels left (hv_conv3x3 / hv_gemm gn_part) ---------
// grid (ceil(groups / 4), n_images), one wavefront per (image, group).  The group's items -- (part, channel) pairs of {sum, sum
// of squares}, from one or two sources (channel concat) -- are spread over the lanes and summed in double precision: the
// partial sums are fp32 over at most a few hundred values each, the merge over up to ~10^4 of them must not lose the digits
// that  var = Q / n - mean^2  cancels.
typedef hv_gn_parts_params HvGnPartsParams;
__global__ __launch_bounds__(256) void hv_gn_from_parts_kernel(HvGnPartsParams p) {
    const int C = p.C1 + p.C2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.x * 4 + wave, img = blockIdx.y;
    const int cg = C / p.groups;
    const bool live = g < p.groups;  // no early return: every lane takes part in the wave shuffles below
    double S = 0.0, Q = 0.0;
    if (live) {
        const int c_lo = g * cg, c_hi = c_lo + cg;
        // source 1: channels [c_lo, min(c_hi, C1)), source 2: the rest
        const int a1 = min(c_lo, p.C1), b1 = min(c_hi, p.C1);
        const int n1 = (b1 - a1) * p.parts1;
        for (int i = lane; i < n1; i += 64) {
            const int part = i / (b1 - a1), c = a1 + i % (b1 - a1);
            const float* src = p.part1 + (((long)img * p.parts1 + part) * p.C1 + c) * 2;
            S += (double)src[0];
            Q += (double)src[1];
        }
        const int a2 = max(c_lo, p.C1) - p.C1, b2 = max(c_hi, p.C1) - p.C1;
        const int n2 = (b2 - a2) * p.parts2;
        for (int i = lane; i < n2; i += 64) {
            const int part = i / (b2 - a2), c = a2 + i % (b2 - a2);
            const float* src = p.part2 + (((long)img * p.parts2 + part) * p.C2 + c) * 2;
            S += (double)src[0];
            Q += (double)src[1];
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        S += __shfl_xor(S, o);
        Q += __shfl_xor(Q, o);
    }
    const double n = (double)cg * (double)p.pixels;
    const double mean = S / n;
    double var = Q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = 1.0f / sqrtf((float)var + p.eps);
    if (!live) return;
    for (int c = g * cg + lane; c < (g + 1) * cg; c += 64) {
        const float sc = rstd * p.gamma[c];
        p.scale[(long)img * C + c] = sc;
        p.shift[(long)img * C + c] = p.beta[c] - (float)mean * sc;
    }
}

static inline int hv_gn_from_parts_launch(const HvGnPartsParams& p, hipStream_t stream) {
    const int C = p.C1 + p.C2;
    if (p.groups <= 0 || p.n_images <= 0) return -1;
    if (p.C1 <= 0 || p.C2 < 0 || C % p.groups != 0 || p.parts1 <= 0 || p.part1 == nullptr || p.pixels <= 0) return -1;
    if (p.C2 > 0 && (p.part2 == nullptr || p.parts2 <= 0)) return -1;
    hv_note("hv_gn_from_parts_kernel | n=%d C=%d parts=%d+%d", p.n_images, C, p.parts1, p.C2 > 0 ? p.parts2 : 0);
    hv_launch(hv_gn_from_parts_kernel, dim3((p.groups + 3) / 4, p.n_images), dim3(256), stream, p);
    return 0;
}

// ---- LayerNorm mean / rstd from the partial row sums the producing GEMM left (ln_part [M][parts][2]): one thread per row
__global__ __launch_bounds__(256) void hv_ln_from_parts_kernel(const float* part, int parts, int M, int C, float eps, float* mean_out,
                                                               float* rstd_out) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float* src = part + (long)m * parts * 2;
    float s = 0.f, q = 0.f;
    for (int i = 0; i < parts; ++i) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(src + 2 * i);
        s += v[0];
        q += v[1];
    }
    const float mean = s / (float)C;
    const float var = fmaxf(q / (float)C - mean * mean, 0.f);
    mean_out[m] = mean;
    rstd_out[m] = 1.0f / sqrtf(var + eps);
}
static inline int hv_ln_from_parts_launch(const float* part, int parts, int M, int C, float eps, float* mean, float* rstd,
                                          hipStream_t stream) {
    if (parts <= 0 || M <= 0 || C <= 0) return -1;  // (parts of any width: 64 columns from the tile kernels, 80 from hv_gemm_wr_kernel)
    hv_note("hv_ln_from_parts_kernel | M=%d C=%d", M, C);
    hv_launch(hv_ln_from_parts_kernel, dim3((M + 255) / 256), dim3(256), stream, part, parts, M, C, eps, mean, rstd);
    return 0;
}

// ---- LayerNorm statistics: one wavefront per row -----------------------------------------------
#define HV_LN_MAXV 4  // 16-byte vectors per lane: C <= 64 * 8 * 4 = 2048

__global__ __launch_bounds__(256) void hv_ln_stats_kernel(const bf16_t* X, long ldx, int M, int C, float eps,
                                                          float* mean_out, float* rstd_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    const int CV = C / 8;
    // no early return: every lane takes part in the wave shuffles below
    const bool live = row < M;
    float f[HV_LN_MAXV][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < HV_LN_MAXV; ++i) {
        const int cv = lane + 64 * i;
        const bool on = live && cv < CV;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (on) v = hv_ld16(X + (long)row * ldx + cv * 8);
        hv_unpack8(v, f[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += f[i][e];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < HV_LN_MAXV; ++i) {
        const int cv = lane + 64 * i;
        if (cv < CV) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = f[i][e] - mean;
                q += d * d;
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o);
    if (live && lane == 0) {
        mean_out[row] = mean;
        rstd_out[row] = 1.0f / sqrtf(q / (float)C + eps);
    }
}

// Several rows per wavefront for the widths of this model (C = 320 / 640 / 1280 = 5 x 16-byte vectors on LPR = 8 / 16 / 32
// lanes): the one-row-per-wave kernel above keeps 40 of 64 lanes busy at C = 320 with ONE load each and spends twelve
// full-wave shuffle steps per 640-byte row (2.4 TB/s); here every lane has five loads in flight, a row is reduced over its
// LPR lanes only, and an instruction still covers whole 128-byte lines (the LPR lanes of a row read consecutive chunks).
template <int LPR>
__global__ __launch_bounds__(256) void hv_ln_stats_rows_kernel(const bf16_t* X, long ldx, int M, float eps, float* mean_out,
                                                               float* rstd_out) {
    constexpr int RPW = 64 / LPR, C = LPR * 5 * 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = (blockIdx.x * 4 + wave) * RPW + lane / LPR, j = lane % LPR;
    const bool live = row < M;  // no early return: every lane takes part in the shuffles below
    float f[5][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (live) v = hv_ld16(X + (long)row * ldx + (j + LPR * i) * 8);
        hv_unpack8(v, f[i]);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += f[i][e];
#pragma unroll
    for (int o = LPR / 2; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = f[i][e] - mean;
            q += d * d;
        }
#pragma unroll
    for (int o = LPR / 2; o >= 1; o >>= 1) q += __shfl_xor(q, o);
    if (live && j == 0) {
        mean_out[row] = mean;
        rstd_out[row] = 1.0f / sqrtf(q / (float)C + eps);
    }
}

static inline void hv_layernorm_launch(const bf16_t* X, long ldx, int M, int C, float eps, float* mean, float* rstd,
                                       hipStream_t stream) {
    hv_note("hv_ln_stats_kernel | M=%d C=%d", M, C);
    if (C == 320)
        hv_launch(hv_ln_stats_rows_kernel<8>, dim3((M + 31) / 32), dim3(256), stream, X, ldx, M, eps, mean, rstd);
    else if (C == 640)
        hv_launch(hv_ln_stats_rows_kernel<16>, dim3((M + 15) / 16), dim3(256), stream, X, ldx, M, eps, mean, rstd);
    else if (C == 1280)
        hv_launch(hv_ln_stats_rows_kernel<32>, dim3((M + 7) / 8), dim3(256), stream, X, ldx, M, eps, mean, rstd);
    else
        hv_launch(hv_ln_stats_kernel, dim3((M + 3) / 4), dim3(256), stream, X, ldx, M, C, eps, mean, rstd);
}
