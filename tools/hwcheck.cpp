// Seconds-long hardware check through the C ABI alone (no Python, no torch: runs in the last seconds of a GPU budget):
//  1. hv_groupnorm_affine (partial sums + the wave-merge finalize) against a double-precision host reference,
//  2. hv_gemm under the kernel selections of HV_TUNE_GEMM_GLDS (1 default / 2 = 256x256x64 / 3 = 128x128x64 / 0 = register
//     staged) against each other (bit for bit) and against sampled host rows, at shapes of the denoising step,
//  3. the 3x3 convolution with 64-channel reduction chunks against the 32-channel kernel.
// Option: --bench prints ms per launch of the step's GEMM shapes under selections 1 / 2 / 3 and of the convolution under both
// chunk sizes (HIP events, no Python start-up).
// Build: hipcc -O2 -Wno-unused-value tools/hwcheck.cpp -Iinclude -Lhumanvid_amd/lib -lhumanvid_hip -Wl,-rpath,'$ORIGIN/../../humanvid_amd/lib' -o tools/bin/hwcheck
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "humanvid_hip.h"

static uint32_t g_s = 12345u;
static inline float rnd() {  // uniform in [-1, 1)
    g_s = g_s * 1664525u + 1013904223u;
    return (float)(int32_t)g_s * (1.0f / 2147483648.0f);
}
static inline uint16_t f2b(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float b2f(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
template <class T>
static T* dev(const std::vector<T>& h) {
    T* d;
    hipMalloc(&d, h.size() * sizeof(T));
    hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}
template <class T>
static T* devz(size_t n) {
    T* d;
    hipMalloc(&d, n * sizeof(T));
    hipMemset(d, 0, n * sizeof(T));
    return d;
}

static int check_gn(int n, int pixels, int C1, int C2, int splits, float offset, const std::vector<int>& imgs) {
    const int C = C1 + C2, groups = 32, cg = C / groups;
    std::vector<uint16_t> x1((size_t)n * pixels * C1), x2((size_t)n * pixels * (C2 ? C2 : 1));
    for (auto& v : x1) v = f2b(rnd() * 1.5f + offset);
    for (auto& v : x2) v = f2b(rnd() * 0.7f - offset);
    std::vector<float> gamma(C), beta(C);
    for (int c = 0; c < C; ++c) gamma[c] = 1.f + 0.2f * rnd(), beta[c] = 0.1f * rnd();
    hv_groupnorm_params p;
    memset(&p, 0, sizeof p);
    uint16_t *dx1 = dev(x1), *dx2 = dev(x2);
    float *dg = dev(gamma), *db = dev(beta);
    p.X = dx1, p.C1 = C1, p.X2 = C2 ? dx2 : nullptr, p.C2 = C2;
    p.n_images = n, p.pixels = pixels, p.groups = groups, p.eps = 1e-5f;
    p.gamma = dg, p.beta = db, p.splits = splits;
    p.partial = devz<float>((size_t)n * 64 * groups * 2);
    p.scale = devz<float>((size_t)n * C), p.shift = devz<float>((size_t)n * C);
    const int rc = hv_groupnorm_affine(&p, nullptr);
    hipDeviceSynchronize();
    std::vector<float> sc((size_t)n * C), sh((size_t)n * C);
    hipMemcpy(sc.data(), p.scale, sc.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(sh.data(), p.shift, sh.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int img : imgs)
        for (int g = 0; g < groups; ++g) {
            double s = 0, q = 0;
            for (int px = 0; px < pixels; ++px)
                for (int c = g * cg; c < (g + 1) * cg; ++c) {
                    const double v = c < C1 ? b2f(x1[((size_t)img * pixels + px) * C1 + c])
                                            : b2f(x2[((size_t)img * pixels + px) * C2 + (c - C1)]);
                    s += v, q += v * v;
                }
            const double cnt = (double)pixels * cg, mean = s / cnt, var = q / cnt - mean * mean;
            const double rstd = 1.0 / sqrt(var + 1e-5);
            for (int c = g * cg; c < (g + 1) * cg; ++c) {
                const double rs = rstd * gamma[c], rh = beta[c] - mean * rs;
                // error of the normalised value at a typical input x = mean + 1/rstd
                const double xv = mean + 1.0 / rstd;
                const double e = fabs((xv * sc[(size_t)img * C + c] + sh[(size_t)img * C + c]) - (xv * rs + rh));
                if (e > worst) worst = e;
            }
        }
    printf("groupnorm n=%d pixels=%d C=%d+%d splits=%d: rc=%d max |err| of the normalised value %.3e %s\n", n, pixels, C1, C2,
           splits, rc, worst, (rc == 0 && worst < 2e-4) ? "OK" : "FAIL");
    fflush(stdout);
    return !(rc == 0 && worst < 2e-4);
}

static int check_gemm(int M, int N, int K, int form, int pol_a = 2, int pol_b = 1) {  // form 0: bias + residual in place; 1: LayerNorm fold + V^T tail
    std::vector<uint16_t> x((size_t)M * K), w((size_t)N * K), res((size_t)M * N);
    const float ws = 1.0f / sqrtf((float)K);
    for (auto& v : x) v = f2b(rnd() + 0.25f);
    for (auto& v : w) v = f2b(rnd() * ws * 1.7f);
    for (auto& v : res) v = f2b(rnd());
    std::vector<float> bias(N), mean(M), rstd(M), colsum(N);
    for (auto& v : bias) v = 0.1f * rnd();
    for (auto& v : mean) v = 0.25f + 0.05f * rnd();
    for (auto& v : rstd) v = 1.5f + 0.2f * rnd();
    for (int n = 0; n < N; ++n) {
        double s = 0;
        for (int k = 0; k < K; ++k) s += b2f(w[(size_t)n * K + k]);
        colsum[n] = (float)s;
    }
    const int ns = form == 1 ? (2 * N / 3) : N;
    uint16_t *dx = dev(x), *dw = dev(w);
    float *dbias = dev(bias), *dmean = dev(mean), *drstd = dev(rstd), *dcs = dev(colsum);
    std::vector<uint16_t> out[2], outt[2];
    for (int pol = 0; pol < 2; ++pol) {
        hv_set_tuning(HV_TUNE_GEMM_GLDS, pol ? pol_b : pol_a);
        uint16_t* dy = dev(res);  // the residual stream, updated in place (form 0)
        uint16_t* dyt = devz<uint16_t>((size_t)(N - ns + 1) * M);
        hv_gemm_params p;
        memset(&p, 0, sizeof p);
        p.X = dx, p.ldx = K, p.K1 = 0, p.W = dw, p.Y = dy, p.ldy = form == 1 ? ns : N, p.M = M, p.N = N, p.K = K;
        p.n_split = 0, p.bias = dbias, p.rows_per_image = 1, p.pe_period = 1, p.pe_frames = 1, p.rowvec_period = 1;
        if (form == 0) p.residual = dy, p.ldr = N;
        else p.row_mean = dmean, p.row_rstd = drstd, p.colsum = dcs, p.Yt = dyt, p.ldyt = M, p.n_split = ns;
        const int rc = hv_gemm(&p, nullptr);
        hipDeviceSynchronize();
        if (rc != 0) {
            printf("gemm M=%d N=%d K=%d form=%d policy %d: rc=%d (%s) FAIL\n", M, N, K, form, pol ? pol_b : pol_a, rc, hv_last_error());
            return 1;
        }
        out[pol].resize((size_t)M * (form == 1 ? ns : N));
        hipMemcpy(out[pol].data(), dy, out[pol].size() * 2, hipMemcpyDeviceToHost);
        outt[pol].resize((size_t)(N - ns) * M);
        if (N > ns) hipMemcpy(outt[pol].data(), dyt, outt[pol].size() * 2, hipMemcpyDeviceToHost);
        hipFree(dy), hipFree(dyt);
    }
    hv_set_tuning(HV_TUNE_GEMM_GLDS, 1);
    double dmax = 0;
    for (size_t i = 0; i < out[0].size(); ++i) dmax = fmax(dmax, fabs(b2f(out[0][i]) - b2f(out[1][i])));
    for (size_t i = 0; i < outt[0].size(); ++i) dmax = fmax(dmax, fabs(b2f(outt[0][i]) - b2f(outt[1][i])));
    // sampled host rows against the second policy
    double emax = 0, rms = 0;
    long cnt = 0;
    for (int r = 0; r < 24; ++r) {
        const int m = (int)(((long)r * 7919 + 13) % M);
        for (int n = 0; n < N; ++n) {
            double acc = 0;
            for (int k = 0; k < K; ++k) acc += (double)b2f(x[(size_t)m * K + k]) * b2f(w[(size_t)n * K + k]);
            double ref = form == 0 ? acc + bias[n] + b2f(res[(size_t)m * N + n]) : rstd[m] * (acc - mean[m] * colsum[n]) + bias[n];
            const float got = n < ns ? b2f(out[1][(size_t)m * (form == 1 ? ns : N) + n]) : b2f(outt[1][(size_t)(n - ns) * M + m]);
            emax = fmax(emax, fabs(got - ref)), rms += ref * ref, ++cnt;
        }
    }
    rms = sqrt(rms / cnt);
    const bool ok = dmax <= 0.02 * rms && emax <= 0.02 * rms;
    printf("gemm M=%d N=%d K=%d form=%d: policy %d vs %d max |diff| %.3e, vs host rows max |err| %.3e (rms %.3f) %s\n", M, N, K, form,
           pol_b, pol_a, dmax, emax, rms, ok ? "OK" : "FAIL");
    fflush(stdout);
    hipFree(dx), hipFree(dw);
    return !ok;
}

// --bench: ms per launch of the step's GEMM shapes under the tile policies (operand values do not matter for the timing;
// LayerNorm-fold forms get zero statistics).  Seconds on the box, no Python start-up: the first measurement of a session.
static void bench_gemm(int M, int N, int K, int form, const char* what) {  // form 0 residual, 1 LN + V^T tail, 2 LN + GEGLU, 3 LN
    uint16_t *dx = devz<uint16_t>((size_t)M * K), *dw = devz<uint16_t>((size_t)N * K), *dy = devz<uint16_t>((size_t)M * N);
    uint16_t* dyt = devz<uint16_t>((size_t)N * M / 3 + 64);
    float *dbias = devz<float>(N), *dmean = devz<float>(M), *drstd = devz<float>(M), *dcs = devz<float>(N);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    printf("%-34s M=%6d N=%5d K=%4d:", what, M, N, K);
    for (int pol : {1, 2, 3}) {
        hv_set_tuning(HV_TUNE_GEMM_GLDS, pol);
        hv_gemm_params p;
        memset(&p, 0, sizeof p);
        p.X = dx, p.ldx = K, p.W = dw, p.Y = dy, p.M = M, p.N = N, p.K = K, p.bias = dbias;
        p.rows_per_image = 1, p.pe_period = 1, p.pe_frames = 1, p.rowvec_period = 1;
        p.ldy = form == 2 ? N / 2 : (form == 1 ? 2 * N / 3 : N);
        if (form == 0) p.residual = dy, p.ldr = N;
        else p.row_mean = dmean, p.row_rstd = drstd, p.colsum = dcs;
        if (form == 1) p.Yt = dyt, p.ldyt = M, p.n_split = 2 * N / 3;
        if (form == 2) p.geglu = 1;
        float ms = -1.f;
        int rc = 0;
        for (int rep = 0; rep < 12 && rc == 0; ++rep) {
            if (rep == 2) hipEventRecord(e0);
            rc = hv_gemm(&p, nullptr);
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rc != 0) printf("  p%d rc=%d", pol, rc);
        else printf("  p%d %.4f ms %6.0f TF/s", pol, ms / 10, 2.0 * M * N * K / (ms / 10) / 1e9);
    }
    hv_set_tuning(HV_TUNE_GEMM_GLDS, 1);
    printf("\n");
    fflush(stdout);
    hipFree(dx), hipFree(dw), hipFree(dy), hipFree(dyt);
}

// 3x3 convolution (stride 1, GroupNorm + SiLU prologue, bias, residual) under two values of HV_TUNE_CONV_BIG: outputs compared
// (different summation order: not bit-identical) and both timed.
static int conv_ab(int n, int H, int W, int Cin, int Cout, int tune_a, int tune_b, bool time_only) {
    std::vector<uint16_t> x((size_t)n * H * W * Cin), w((size_t)Cout * 9 * Cin), res((size_t)n * H * W * Cout);
    const float ws = 1.0f / sqrtf(9.0f * Cin);
    for (auto& v : x) v = f2b(rnd());
    for (auto& v : w) v = f2b(rnd() * ws * 1.7f);
    for (auto& v : res) v = f2b(rnd());
    std::vector<float> sc((size_t)n * Cin), sh((size_t)n * Cin), bias(Cout);
    for (auto& v : sc) v = 1.f + 0.3f * rnd();
    for (auto& v : sh) v = 0.2f * rnd();
    for (auto& v : bias) v = 0.1f * rnd();
    uint16_t *dx = dev(x), *dw = dev(w), *dres = dev(res);
    float *dsc = dev(sc), *dsh = dev(sh), *db = dev(bias);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    std::vector<uint16_t> out[2];
    float ms[2] = {0, 0};
    int bad = 0;
    for (int v = 0; v < 2; ++v) {
        hv_set_tuning(HV_TUNE_CONV_BIG, v ? tune_b : tune_a);
        uint16_t* dy = devz<uint16_t>(res.size());
        hv_conv3x3_params p;
        memset(&p, 0, sizeof p);
        p.X = dx, p.C1 = Cin, p.W = dw, p.Y = dy, p.n_images = n, p.Hs = H, p.Ws = W, p.Ho = H, p.Wo = W, p.Cout = Cout;
        p.mode = HV_CONV_S1, p.pro_scale = dsc, p.pro_shift = dsh, p.pro_act = HV_ACT_SILU, p.bias = db;
        p.images_per_rowvec = 1, p.rowvec_ld = Cout, p.residual = dres, p.residual_images = n;
        int rc = 0;
        for (int rep = 0; rep < 7 && rc == 0; ++rep) {
            if (rep == 2) hipEventRecord(e0);
            rc = hv_conv3x3(&p, nullptr);
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[v], e0, e1);
        ms[v] /= 5;
        if (rc != 0) {
            printf("conv tune %d: rc=%d (%s) FAIL\n", v ? tune_b : tune_a, rc, hv_last_error());
            bad = 1;
        }
        out[v].resize(res.size());
        hipMemcpy(out[v].data(), dy, res.size() * 2, hipMemcpyDeviceToHost);
        hipFree(dy);
    }
    hv_set_tuning(HV_TUNE_CONV_BIG, 1);
    double d2 = 0, r2 = 0;
    for (size_t i = 0; i < out[0].size(); ++i) {
        const double a = b2f(out[0][i]), b = b2f(out[1][i]);
        d2 += (a - b) * (a - b), r2 += a * a;
    }
    const double rel = sqrt(d2 / (r2 + 1e-30));
    const bool ok = !bad && rel < 3e-3 && r2 > 0;
    const double tf = 2.0 * 9 * Cin * Cout * (double)n * H * W / 1e9;
    printf("conv n=%d %dx%d Cin=%d Cout=%d: tune %d %.4f ms %5.0f TF/s | tune %d %.4f ms %5.0f TF/s | rel rms diff %.2e %s\n", n, H, W, Cin,
           Cout, tune_a, ms[0], tf / ms[0], tune_b, ms[1], tf / ms[1], rel, (ok || time_only) ? "OK" : "FAIL");
    fflush(stdout);
    hipFree(dx), hipFree(dw), hipFree(dres);
    return (ok || time_only) ? 0 : 1;
}

int main(int argc, char** argv) {
    bool bench = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--bench")) bench = true;
    }
    int bad = 0;
    bad += check_gn(48, 6144, 320, 0, 64, 0.7f, {0, 47});
    bad += check_gn(4, 384, 1280, 640, 8, -2.0f, {0, 3});
    bad += check_gn(2, 9, 640, 0, 4, 0.3f, {0, 1});
    bad += check_gn(3, 96, 2560, 0, 2, 5.0f, {1});
    // selection 1 (default: the fill test sends these level-2 / level-3 shapes to the 128x128x64 kernel) against 2
    // (256x256x64 wherever legal) and against sampled host rows: bit-identical by construction
    bad += check_gemm(18432, 1280, 1280, 0, 2, 1);
    bad += check_gemm(4608, 1280, 1280, 0, 2, 1);
    bad += check_gemm(18432, 3840, 1280, 1, 2, 1);
    bad += check_gemm(36864, 960, 320, 1, 3, 1);    // the streamed level-0 projection: 128x128x64 against the default 256x256x64
    bad += check_gemm(4608, 2560, 64, 0, 3, 2);     // one k-step per tile: an epilogue after every step
    bad += check_gemm(36864, 320, 1280, 0, 0, 1);   // register-staged kernel against the LDS-DMA kernel
    {  // 64-channel reduction chunks (HV_TUNE_CONV_BIG = 3 forces them) against the 32-channel kernel (0)
        bad += conv_ab(4, 96, 64, 320, 320, 0, 3, false);
        bad += conv_ab(3, 21, 13, 128, 200, 0, 3, false);  // ragged patches and output channels
        bad += conv_ab(6, 12, 8, 1280, 1280, 0, 3, false);  // narrow images: 8 x 16 patches
    }
    if (bench) {
        conv_ab(48, 96, 64, 320, 320, 0, 3, true);
        conv_ab(48, 48, 32, 640, 640, 0, 3, true);
        conv_ab(48, 24, 16, 1280, 1280, 0, 3, true);
        conv_ab(48, 12, 8, 1280, 1280, 0, 3, true);
        conv_ab(48, 96, 64, 640, 320, 0, 3, true);
        const int M0 = 48 * 6144, M1 = 48 * 1536, M2 = 48 * 384, M3 = 48 * 96;
        bench_gemm(M0, 960, 320, 1, "level-0 spatial QKV (LN, V^T)");
        bench_gemm(M0, 960, 320, 3, "level-0 temporal QKV (LN)");
        bench_gemm(M0, 2560, 320, 2, "level-0 ff1 (LN, GEGLU)");
        bench_gemm(M0, 320, 320, 0, "level-0 out-proj (+res)");
        bench_gemm(M0, 320, 1280, 0, "level-0 ff2 (+res)");
        bench_gemm(M1, 1920, 640, 1, "level-1 spatial QKV");
        bench_gemm(M1, 5120, 640, 2, "level-1 ff1");
        bench_gemm(M1, 640, 640, 0, "level-1 out-proj");
        bench_gemm(M1, 640, 2560, 0, "level-1 ff2");
        bench_gemm(M2, 3840, 1280, 1, "level-2 spatial QKV");
        bench_gemm(M2, 10240, 1280, 2, "level-2 ff1");
        bench_gemm(M2, 1280, 1280, 0, "level-2 out-proj");
        bench_gemm(M2, 1280, 5120, 0, "level-2 ff2");
        bench_gemm(M3, 10240, 1280, 2, "level-3 ff1");
        bench_gemm(M3, 1280, 1280, 0, "level-3 out-proj");
        bench_gemm(M3, 1280, 5120, 0, "level-3 ff2");
    }
    printf("hwcheck: %s\n", bad ? "FAIL" : "all OK");
    return bad;
}
