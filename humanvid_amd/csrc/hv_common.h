// hv_common.h -- shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
//
// Everything here is written for 64-lane wavefronts and the gfx950 MFMA shapes
// (16x16x32 bf16: 8 bf16 per lane for A and B, 4 fp32 accumulators per lane).
// Activations live in HBM as bf16, channels-last: [image][y][x][channel] == [image][token][channel];
// all accumulation is fp32.
#pragma once
#ifndef HV_EMU
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>
#include <stdio.h>

#include <functional>
#include <utility>
#include <vector>
#include "humanvid_hip.h"  // HV_ACT_* and the parameter structs

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define HV_DEV __device__ __forceinline__

HV_DEV float hv_bf2f(bf16_t b) {
    union {
        uint32_t u;
        float f;
    } c;
    c.u = ((uint32_t)b) << 16;
    return c.f;
}

HV_DEV bf16_t hv_f2bf(float f) {  // round-to-nearest-even, NaN preserved
    union {
        uint32_t u;
        float f;
    } c;
    c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)0x7fc0;
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

HV_DEV uint32_t hv_pack2(float lo, float hi) {
#ifndef HV_EMU
    // gfx950 has a packed fp32 -> bf16 (RNE) conversion: one v_cvt_pk_bf16_f32 instead of ~10 integer ops
    typedef __bf16 hv_bf2_t __attribute__((ext_vector_type(2)));
    typedef float hv_f2_t __attribute__((ext_vector_type(2)));
    hv_f2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hv_bf2_t));
#else
    return (uint32_t)hv_f2bf(lo) | ((uint32_t)hv_f2bf(hi) << 16);
#endif
}

// a * b for a, b < 2^24 (row index x row stride in bytes): full-rate v_mul_u32_u24 instead of the quarter-rate v_mul_lo_u32
HV_DEV unsigned hv_umul24(unsigned a, unsigned b) {
#ifndef HV_EMU
    return __umul24(a, b);
#else
    return a * b;
#endif
}

// sum over the 16 lanes of a DPP row (lanes with equal lane >> 4), returned in every lane: four v_add_f32 with DPP operands
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror) -- no LDS round trips (__shfl_xor is ds_bpermute)
HV_DEV float hv_row16_sum(float x) {
#ifndef HV_EMU
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, true));
#else
    x += __shfl_xor(x, 1);
    x += __shfl_xor(x, 2);
    x += __shfl_xor(x, 4);
    x += __shfl_xor(x, 8);
#endif
    return x;
}

HV_DEV float hv_silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504089f * x)); }

HV_DEV float hv_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// 16-byte global / LDS vector access
HV_DEV u32x4 hv_ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
HV_DEV void hv_st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }
HV_DEV u32x2 hv_ld8(const void* p) { return *reinterpret_cast<const u32x2*>(p); }
HV_DEV void hv_st8(void* p, u32x2 v) { *reinterpret_cast<u32x2*>(p) = v; }
// (the write-through / non-temporal forms of the output stores measured no gain in rounds 1-2 and are gone)
HV_DEV void hv_st8_stream(void* p, u32x2 v) { *reinterpret_cast<u32x2*>(p) = v; }

HV_DEV bf16x8 hv_as_bf16x8(u32x4 v) {
    union {
        u32x4 u;
        bf16x8 s;
    } c;
    c.u = v;
    return c.s;
}

HV_DEV void hv_unpack8(u32x4 v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = hv_bf2f((bf16_t)(v[i] & 0xffffu));
        f[2 * i + 1] = hv_bf2f((bf16_t)(v[i] >> 16));
    }
}

HV_DEV u32x4 hv_pack8(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = hv_pack2(f[2 * i], f[2 * i + 1]);
    return v;
}


HV_DEV float hv_act(float x, int act) {
    if (act == HV_ACT_SILU) return hv_silu(x);
    if (act == HV_ACT_RELU) return x > 0.f ? x : 0.f;
    return x;
}

// ---- asynchronous global -> LDS copies (LDS-DMA) --------------------------------------------
// One wave-instruction moves 64 x 16 B: every lane supplies its own global address, the LDS
// destination is wave-uniform base + lane * 16 (so any LDS swizzle is applied on the SOURCE side).
// Completion is tracked by vmcnt; hv_vm_wait<N>() leaves at most N such loads in flight.
#ifndef HV_EMU
HV_DEV void hv_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// Same copy issued from inline asm, with the source as a wave-uniform base (SGPR pair) + a 32-bit per-lane byte offset and
// the LDS destination as a scalar byte address.  hipcc's wait-count tracker does not see it: with the builtin form it makes
// every ds_read that follows an LDS-DMA wait vmcnt(0) ("the DMA may alias the read") -- seen in the convolution's k-loop,
// where it drained the ring in every step -- whereas here completion is tracked by hand anyway (hv_vm_wait).  Ordinary loads
// stay correct: vmcnt retires in order, so a compiler-computed count that ignores these copies can only wait longer.
HV_DEV void hv_glds16_s(const void* base_uniform, unsigned byte_ofs, void* lds_wave_base) {
    const unsigned lds_addr_uniform = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)lds_wave_base;
    // M0 is compiler-reserved and not preserved around an asm statement: save / restore it inside the statement
    // (cdna_hip_programming.md 5.7).  Leading s_nop 4: an SGPR operand fresh from v_readfirstlane read as a VMEM base.
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(byte_ofs), "s"(base_uniform), "s"(lds_addr_uniform)
                 : "memory");
}
template <int N>
HV_DEV void hv_vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
HV_DEV void hv_barrier_raw() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}
#else
HV_DEV void hv_glds16(const void* gsrc, void* lds_wave_base) {
    memcpy((char*)lds_wave_base + (threadIdx.x & 63) * 16, gsrc, 16);  // emulator: synchronous
}
HV_DEV void hv_glds16_s(const void* base_uniform, unsigned byte_ofs, void* lds_wave_base) {
    memcpy((char*)lds_wave_base + (threadIdx.x & 63) * 16, (const char*)base_uniform + byte_ofs, 16);
}
template <int N>
HV_DEV void hv_vm_wait() {}
HV_DEV void hv_barrier_raw() { __syncthreads(); }
#endif

// ---- gfx950 cross-lane / transposing LDS primitives -----------------------------------------------------
// hv_lds_tr4: ds_read_b64_tr_b16.  Every lane supplies the 8-byte aligned LDS address of 4 consecutive bf16; within each
// 16-lane group the 16 x 4 values are transposed: lane t of the group receives element (t & 3) of the lanes 4e + (t >> 2),
// e = 0..3.  With lane t pointing at [row t >> 2][columns 4 (t & 3) ..] of a row-major block, lane t ends up with
// column t of rows 0..3 -- a row-major V tile is read as V^T MFMA fragments.  (tools/tr_probe.hip prints the map.)
// hv_swap32: the value of the same lane index in the other 32-lane half (v_permlane32_swap).
// The builtin form makes hipcc (ROCm 7.2) drain vmcnt(0) in front of the read whenever an LDS-DMA is in flight, so the
// kernels use the inline-asm form: issue a batch with hv_lds_tr4_issue(), then hv_lds_tr4_wait() once before the first
// consumer (the compiler does not count these reads: cdna_hip_programming.md 5.7 form (iii)).
#ifndef HV_EMU
HV_DEV bf16x4 hv_lds_tr4(const void* lds_ptr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)lds_ptr);
}
HV_DEV void hv_lds_tr4_issue(bf16x4& dst, const void* lds_ptr) {
    const unsigned a = (unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)lds_ptr;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(a));
}
template <int OFF>
HV_DEV void hv_lds_tr4_issue_off(bf16x4& dst, unsigned lds_addr) {  // base VGPR + immediate offset (< 65536)
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "i"(OFF));
}
HV_DEV unsigned hv_lds_addr(const void* lds_ptr) {
    return (unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)lds_ptr;
}
HV_DEV void hv_lds_tr4_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
HV_DEV float hv_swap32(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    // r[0]: lanes 32-63 now hold the lower half's values; r[1]: lanes 0-31 hold the upper half's
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}
#endif

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}) -- for loop bodies
// that need their index as a constant expression (immediate operands of inline asm)
template <int... I, class F>
HV_DEV void hv_static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
HV_DEV void hv_static_for(F&& f) {
    hv_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// ---- launch plumbing ----------------------------------------------------------------------
// One launch helper for both builds: the real one uses the HIP triple-chevron launch on the
// caller's stream, the emulator (tests only) runs the workgroups on host fibers.
// Command lists: while a list is being recorded every launch is also appended to it as a closure
// (kernel, geometry, by-value arguments); hv_cmdlist_run() re-issues the closures on a stream with
// one native loop -- the host-side alternative to a HIP graph for launch sequences that are
// interleaved with RCCL collectives (frame-sharded runs).
struct HvCmdList {
    std::vector<std::function<void(hipStream_t)>> cmds;
#ifndef HV_EMU
    // hv_cmdlist_run: the closures are captured once into a HIP graph (on a capture stream of their own, so that the caller's
    // stream may be the null stream) and every run is one hipGraphLaunch on the caller's stream -- a frame-sharded step is
    // ~700 launches cut into ~90 segments by its collectives: ~90 graph launches instead of ~700 std::function calls + kernel
    // launches from the host per step and rank
    hipGraphExec_t exec = nullptr;
    bool no_graph = false;  // the capture of this list failed once: re-issue its closures instead
    ~HvCmdList() {
        if (exec) (void)hipGraphExecDestroy(exec);
    }
#endif
};
extern thread_local HvCmdList* g_hv_recording;  // defined in hv_api.cpp

// Launch profile (hv_profile_begin / hv_profile_end): while a profile is open every launch is bracketed by two HIP
// events on its own stream and filed under the note its launcher left with hv_note() -- "kernel variant | shape".
// bench.py prices the notes (flops / algorithmic bytes from the shape) to build the roofline from the step's REAL
// launches (real epilogues, multiplicities and variants) instead of a synthetic replay.
struct HvProfEntry {
    char key[192];
#ifndef HV_EMU
    hipEvent_t e0, e1;
#endif
};
struct HvProfile {
    std::vector<HvProfEntry> entries;
};
extern thread_local HvProfile* g_hv_prof;  // defined in hv_api.cpp
extern thread_local char g_hv_note[192];
#define hv_note(...)                                                        \
    do {                                                                    \
        if (g_hv_prof) snprintf(g_hv_note, sizeof(g_hv_note), __VA_ARGS__); \
    } while (0)

template <class... KArgs, class... Args>
static inline void hv_launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, hipStream_t stream, Args... args) {
#ifdef HV_EMU
    (void)stream;
    if (g_hv_recording)
        g_hv_recording->cmds.push_back([=](hipStream_t) { hvemu::launch(grid, block, [&]() { kernel(args...); }); });
    hvemu::launch(grid, block, [&]() { kernel(args...); });
#else
    if (g_hv_recording)
        g_hv_recording->cmds.push_back([=](hipStream_t s) { kernel<<<grid, block, 0, s>>>(args...); });
    if (g_hv_prof) {
        HvProfEntry en;
        snprintf(en.key, sizeof(en.key), "%s", g_hv_note[0] ? g_hv_note : "other");
        g_hv_note[0] = 0;
        (void)hipEventCreate(&en.e0);
        (void)hipEventCreate(&en.e1);
        (void)hipEventRecord(en.e0, stream);
        kernel<<<grid, block, 0, stream>>>(args...);
        (void)hipEventRecord(en.e1, stream);
        g_hv_prof->entries.push_back(en);
        return;
    }
    kernel<<<grid, block, 0, stream>>>(args...);
#endif
}
