// hv_emu.h -- a tiny single-threaded emulator of the HIP execution model (workgroups of
// 64-lane waves, LDS, barriers, wave-collective MFMA / shuffles), built on ucontext fibers.
//
// DEVELOPMENT / TEST INFRASTRUCTURE ONLY.  The build container has no GPU, so the kernels in
// humanvid_amd/csrc are additionally compiled for the host against this header (-DHV_EMU) and
// their index arithmetic (MFMA fragment layouts, LDS tiling, halo handling, online softmax)
// is checked against the oracle on tiny shapes by tests/test_emu_kernels.py.  The product
// library (libhumanvid_hip.so) never contains or loads any of this; nothing here is a
// "CPU fallback".
//
// Fidelity notes:
//  * MFMA C/D layouts follow /opt/skills/guides/cdna_hip_programming.md section 3:
//      16x16:  col = lane & 15, row = (lane >> 4) * 4 + reg
//      32x32:  col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
//    A/B operands: lane (i = lane & 15|31, g = lane >> 4|5) holds k = g * KV + 0..KV-1 where KV
//    is the per-lane vector length (8 for the gfx950 double-K shapes, 4 for 16x16x16).
//  * products are accumulated in fp32 in k order, inputs are bf16.
//  * fibers run to the next barrier/collective in thread order, so a missing __syncthreads()
//    shows up as a wrong result rather than being hidden by lock-step execution.
#pragma once
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

namespace hvemu {

struct uint3_ {
    unsigned x, y, z;
};
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct WaveState {
    int arrived = 0;
    int gen = 0;
    alignas(16) unsigned char in[64][160];
    alignas(16) unsigned char out[64][64];
};

struct Fiber {
    ucontext_t ctx;
    uint3_ tid;
    int lane, wave;
    bool done = false;
    std::vector<char> stack;
};

struct Block {
    std::vector<Fiber> fibers;
    std::vector<WaveState> waves;
    int bar_arrived = 0, bar_gen = 0, alive = 0;
};

inline Block* g_block = nullptr;
inline Fiber* cur = nullptr;
inline ucontext_t g_sched;
inline uint3_ g_blockIdx, g_blockDim, g_gridDim;
inline std::function<void()>* g_body = nullptr;

inline void yield_() { swapcontext(&cur->ctx, &g_sched); }

inline void fiber_entry() {
    (*g_body)();
    cur->done = true;
    g_block->alive--;
    // a thread that exits early must not dead-lock a barrier the others are waiting on
    swapcontext(&cur->ctx, &g_sched);
}

inline void syncthreads() {
    Block& b = *g_block;
    int my = b.bar_gen;
    if (++b.bar_arrived >= b.alive) {
        b.bar_arrived = 0;
        b.bar_gen++;
        return;
    }
    while (b.bar_gen == my) yield_();
}

// run `compute(in[64], out[64])` once per wave when all 64 lanes have arrived
template <class F>
inline void wave_collective(const void* in, size_t in_sz, void* out, size_t out_sz, F compute) {
    WaveState& w = g_block->waves[cur->wave];
    if (in_sz > sizeof(w.in[0]) || out_sz > sizeof(w.out[0])) {
        fprintf(stderr, "hvemu: collective payload too large\n");
        abort();
    }
    memcpy(w.in[cur->lane], in, in_sz);
    int my = w.gen;
    if (++w.arrived == 64) {
        compute(w);
        w.arrived = 0;
        w.gen++;
    } else {
        while (w.gen == my) yield_();
    }
    memcpy(out, w.out[cur->lane], out_sz);
}

inline void launch(dim3 grid, dim3 block, std::function<void()> body, size_t stack_bytes = 256 * 1024) {
    const int nthreads = block.x * block.y * block.z;
    if (nthreads % 64) {
        fprintf(stderr, "hvemu: block size must be a multiple of 64\n");
        abort();
    }
    g_body = &body;
    g_blockDim = {block.x, block.y, block.z};
    g_gridDim = {grid.x, grid.y, grid.z};
    Block blk;
    blk.fibers.resize(nthreads);
    blk.waves.resize(nthreads / 64);
    for (auto& f : blk.fibers) f.stack.resize(stack_bytes);
    g_block = &blk;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g_blockIdx = {bx, by, bz};
                blk.bar_arrived = 0;
                blk.alive = nthreads;
                for (auto& w : blk.waves) w.arrived = 0;
                for (int t = 0; t < nthreads; ++t) {
                    Fiber& f = blk.fibers[t];
                    f.done = false;
                    f.tid = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
                    f.lane = t % 64;
                    f.wave = t / 64;
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack.data();
                    f.ctx.uc_stack.ss_size = f.stack.size();
                    f.ctx.uc_link = &g_sched;
                    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
                }
                long guard = 0;
                while (blk.alive > 0) {
                    for (int t = 0; t < nthreads; ++t) {
                        Fiber& f = blk.fibers[t];
                        if (f.done) continue;
                        cur = &f;
                        swapcontext(&g_sched, &f.ctx);
                    }
                    if (++guard > 50000000L) {
                        fprintf(stderr, "hvemu: dead-lock (divergent barrier?)\n");
                        abort();
                    }
                }
            }
    g_block = nullptr;
    cur = nullptr;
}

}  // namespace hvemu

// ---- the subset of the HIP device language the kernels use ---------------------------------
using dim3 = hvemu::dim3;
#define threadIdx (hvemu::cur->tid)
#define blockIdx (hvemu::g_blockIdx)
#define blockDim (hvemu::g_blockDim)
#define gridDim (hvemu::g_gridDim)
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__
inline void __syncthreads() { hvemu::syncthreads(); }

typedef short hv_s8 __attribute__((ext_vector_type(8)));
typedef short hv_s4 __attribute__((ext_vector_type(4)));
typedef float hv_f4 __attribute__((ext_vector_type(4)));
typedef float hv_f16 __attribute__((ext_vector_type(16)));

namespace hvemu {
inline float bf2f(short s) {
    uint32_t u = ((uint32_t)(uint16_t)s) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

template <int MN, int KV, int NREG, class VA, class VC>
inline VC mfma(VA a, VA b, VC c) {
    struct In {
        VA a, b;
        VC c;
    } in{a, b, c};
    VC out;
    wave_collective(&in, sizeof(in), &out, sizeof(out), [](WaveState& w) {
        constexpr int G = 64 / MN;  // lane groups along k
        static float A[MN][G * KV], B[G * KV][MN];
        for (int l = 0; l < 64; ++l) {
            In li;
            memcpy(&li, w.in[l], sizeof(In));
            for (int j = 0; j < KV; ++j) {
                A[l % MN][(l / MN) * KV + j] = bf2f(li.a[j]);
                B[(l / MN) * KV + j][l % MN] = bf2f(li.b[j]);
            }
        }
        for (int l = 0; l < 64; ++l) {
            In li;
            memcpy(&li, w.in[l], sizeof(In));
            VC o;
            for (int r = 0; r < NREG; ++r) {
                int col = l % MN;
                int row = (MN == 16) ? (l / 16) * 4 + r : (r & 3) + 8 * (r >> 2) + 4 * (l / 32);
                float acc = li.c[r];
                for (int k = 0; k < G * KV; ++k) acc = fmaf(A[row][k], B[k][col], acc);
                o[r] = acc;
            }
            memcpy(w.out[l], &o, sizeof(VC));
        }
    });
    return out;
}

template <class T>
inline T shfl_idx(T v, int src_lane_of_me) {
    struct In {
        T v;
        int src;
    } in{v, src_lane_of_me};
    T out;
    wave_collective(&in, sizeof(in), &out, sizeof(out), [](WaveState& w) {
        for (int l = 0; l < 64; ++l) {
            In li;
            memcpy(&li, w.in[l], sizeof(In));
            In ls;
            memcpy(&ls, w.in[li.src & 63], sizeof(In));
            memcpy(w.out[l], &ls.v, sizeof(T));
        }
    });
    return out;
}
}  // namespace hvemu

// ---- OCP e4m3 (fn) as gfx950's v_cvt_pk_fp8_f32 / v_mfma_f32_16x16x32_fp8_fp8 use it: bias 7, 3 mantissa bits, max 448,
//      no infinities; conversion rounds to nearest even and saturates
namespace hvemu {
inline float fp8_to_f(unsigned char b) {
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 0) v = ldexpf((float)m / 8.0f, -6);
    else if (e == 15 && m == 7) v = NAN;
    else v = ldexpf(1.0f + (float)m / 8.0f, e - 7);
    return s ? -v : v;
}
inline unsigned char f_to_fp8(float x) {
    if (x != x) return 0x7f;
    const unsigned char s = std::signbit(x) ? 0x80 : 0;
    float a = fabsf(x);
    if (a >= 448.0f) return s | 0x7e;
    if (a < ldexpf(1.0f, -6)) {  // subnormal: multiples of 2^-9
        const int q = (int)nearbyintf(a * 512.0f);  // FE_TONEAREST: ties to even
        return s | (unsigned char)q;                 // q == 8 is exactly the smallest normal (exp 1, mant 0)
    }
    int e;
    const float fr = frexpf(a, &e);  // a = fr * 2^e, fr in [0.5, 1)
    int q = (int)nearbyintf(fr * 16.0f);  // 8 .. 16
    int ex = e - 1 + 7;
    if (q == 16) q = 8, ++ex;
    if (ex > 15 || (ex == 15 && q - 8 > 6)) return s | 0x7e;
    return s | (unsigned char)((ex << 3) | (q - 8));
}
inline unsigned cvt_pk_fp8(float a, float b, unsigned old, bool hi) {
    const unsigned v = (unsigned)f_to_fp8(a) | ((unsigned)f_to_fp8(b) << 8);
    return hi ? (old & 0x0000ffffu) | (v << 16) : (old & 0xffff0000u) | v;
}
inline hv_f4 mfma_fp8(long a, long b, hv_f4 c) {
    struct In {
        long a, b;
        hv_f4 c;
    } in{a, b, c};
    hv_f4 out;
    wave_collective(&in, sizeof(in), &out, sizeof(out), [](WaveState& w) {
        static float A[16][32], B[32][16];
        for (int l = 0; l < 64; ++l) {
            In li;
            memcpy(&li, w.in[l], sizeof(In));
            for (int j = 0; j < 8; ++j) {
                A[l % 16][(l / 16) * 8 + j] = fp8_to_f((unsigned char)((unsigned long)li.a >> (8 * j)));
                B[(l / 16) * 8 + j][l % 16] = fp8_to_f((unsigned char)((unsigned long)li.b >> (8 * j)));
            }
        }
        for (int l = 0; l < 64; ++l) {
            In li;
            memcpy(&li, w.in[l], sizeof(In));
            hv_f4 o;
            for (int r = 0; r < 4; ++r) {
                const int col = l % 16, row = (l / 16) * 4 + r;
                float acc = li.c[r];
                for (int k = 0; k < 32; ++k) acc = fmaf(A[row][k], B[k][col], acc);
                o[r] = acc;
            }
            memcpy(w.out[l], &o, sizeof(hv_f4));
        }
    });
    return out;
}
}  // namespace hvemu
inline int atomicMax(int* p, int v) {
    const int old = *p;
    if (v > old) *p = v;
    return old;
}

inline hv_f4 __builtin_amdgcn_mfma_f32_16x16x32_bf16(hv_s8 a, hv_s8 b, hv_f4 c, int, int, int) {
    return hvemu::mfma<16, 8, 4>(a, b, c);
}
inline hv_f4 __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(hv_s4 a, hv_s4 b, hv_f4 c, int, int, int) {
    return hvemu::mfma<16, 4, 4>(a, b, c);
}
inline hv_f16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(hv_s8 a, hv_s8 b, hv_f16 c, int, int, int) {
    return hvemu::mfma<32, 8, 16>(a, b, c);
}
// ds_read_b64_tr_b16 / v_permlane32_swap as used through hv_common.h (hardware map: tools/tr_probe.hip)
inline hv_s4 hv_lds_tr4(const void* lds_ptr) {
    hv_s4 mine;
    memcpy(&mine, lds_ptr, 8);
    hv_s4 out;
    hvemu::wave_collective(&mine, sizeof(mine), &out, sizeof(out), [](hvemu::WaveState& w) {
        for (int l = 0; l < 64; ++l) {
            const int gb = l & ~15, t = l & 15;
            hv_s4 o;
            for (int e = 0; e < 4; ++e) {
                hv_s4 src;
                memcpy(&src, w.in[gb + 4 * e + (t >> 2)], sizeof(src));
                o[e] = src[t & 3];
            }
            memcpy(w.out[l], &o, sizeof(o));
        }
    });
    return out;
}
inline void hv_lds_tr4_issue(hv_s4& dst, const void* lds_ptr) { dst = hv_lds_tr4(lds_ptr); }
template <int OFF>
inline void hv_lds_tr4_issue_off(hv_s4& dst, const unsigned char* lds_addr) { dst = hv_lds_tr4(lds_addr + OFF); }
inline const unsigned char* hv_lds_addr(const void* lds_ptr) { return (const unsigned char*)lds_ptr; }
inline void hv_lds_tr4_wait() {}
template <class T>
inline T __shfl_xor(T v, int mask);
inline float hv_swap32(float x) { return __shfl_xor(x, 32); }

template <class T>
inline T __shfl_xor(T v, int mask) {
    return hvemu::shfl_idx(v, hvemu::cur->lane ^ mask);
}
template <class T>
inline T __shfl(T v, int lane) {
    return hvemu::shfl_idx(v, lane);
}
inline int __any(int pred) {
    int out;
    hvemu::wave_collective(&pred, sizeof(pred), &out, sizeof(out), [](hvemu::WaveState& w) {
        int any = 0;
        for (int l = 0; l < 64; ++l) {
            int v;
            memcpy(&v, w.in[l], sizeof(int));
            any |= (v != 0);
        }
        for (int l = 0; l < 64; ++l) memcpy(w.out[l], &any, sizeof(int));
    });
    return out;
}
inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / sqrtf(x); }
inline float atomicAdd(float* p, float v) {
    float o = *p;
    *p = o + v;
    return o;
}

// host runtime stand-ins used by the C-ABI wrappers
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
    memset(p, v, n);
    return 0;
}
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
