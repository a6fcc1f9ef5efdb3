// hv_api.cpp -- the C ABI (include/humanvid_hip.h) over the gfx950 kernels.
// Built by __graft_entry__.build():  hipcc --offload-arch=gfx950 -O3 -shared -fPIC ...
#include "humanvid_hip.h"

#include <stdio.h>
#include <string.h>

#include <cstring>
#include <string>

#include "hv_common.h"
#include "hv_kernels.h"

static thread_local char g_err[512] = "";
thread_local HvCmdList* g_hv_recording = nullptr;
thread_local HvProfile* g_hv_prof = nullptr;
thread_local char g_hv_note[192] = "";

static int hv_fail(int code, const char* what) {
    snprintf(g_err, sizeof(g_err), "%s", what);
    return code;
}

static int hv_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return HV_EHIP;
    }
    return HV_OK;
}

extern "C" {

const char* hv_last_error(void) { return g_err; }
int hv_abi_version(void) { return 2; }  // 2: gn_part fields, hv_groupnorm_from_parts (round 3)

int hv_struct_sizes(int* out, int capacity) {
    const int s[] = {(int)sizeof(hv_gemm_params),      (int)sizeof(hv_conv3x3_params),
                     (int)sizeof(hv_groupnorm_params), (int)sizeof(hv_attention_params),
                     (int)sizeof(hv_temporal_attention_params), (int)sizeof(hv_gn_parts_params)};
    const int n = (int)(sizeof(s) / sizeof(s[0]));
    for (int i = 0; i < n && i < capacity; ++i) out[i] = s[i];
    return n;
}

int hv_gemm(const hv_gemm_params* p, void* stream) {
    if (!p || !p->X || !p->W || !p->Y) return hv_fail(HV_EINVAL, "hv_gemm: null operand");
    if (hvk_gemm(*p, (hipStream_t)stream) != 0)
        return hv_fail(HV_EINVAL, "hv_gemm: need K % 64 == 0, N % 4 == 0 (geglu: N % 32 == 0); gn_part only where hv_gemm_gn_parts() > 0");
    return hv_check_launch("hv_gemm");
}

int hv_conv3x3(const hv_conv3x3_params* p, void* stream) {
    if (!p || !p->X || !p->W || !p->Y) return hv_fail(HV_EINVAL, "hv_conv3x3: null operand");
    if (hvk_conv3x3(*p, (hipStream_t)stream) != 0)
        return hv_fail(HV_EINVAL, "hv_conv3x3: need C1 % 32 == 0, C2 % 32 == 0, Cout % 4 == 0, consistent sizes");
    return hv_check_launch("hv_conv3x3");
}

int hv_groupnorm_affine(const hv_groupnorm_params* p, void* stream) {
    if (!p || !p->X || !p->partial || !p->scale || !p->shift) return hv_fail(HV_EINVAL, "hv_groupnorm: null");
    if (hvk_groupnorm(*p, (hipStream_t)stream) != 0)
        return hv_fail(HV_EINVAL, "hv_groupnorm: need C1 % 8 == 0, C2 % 8 == 0, C % groups == 0, C <= 4096");
    return hv_check_launch("hv_groupnorm_affine");
}

int hv_gemm_gn_parts(const hv_gemm_params* p) { return p ? hvk_gemm_gn_parts(*p) : 0; }
int hv_gemm_ln_parts(const hv_gemm_params* p) { return p ? hvk_gemm_ln_parts(*p) : 0; }
int hv_conv3x3_gn_parts(const hv_conv3x3_params* p) { return p ? hvk_conv3x3_gn_parts(*p) : 0; }
int hv_layernorm_from_parts(const float* part, int parts, int M, int C, float eps, float* mean, float* rstd, void* stream) {
    if (!part || !mean || !rstd) return hv_fail(HV_EINVAL, "hv_layernorm_from_parts: null");
    if (hvk_ln_from_parts(part, parts, M, C, eps, mean, rstd, (hipStream_t)stream) != 0)
        return hv_fail(HV_EINVAL, "hv_layernorm_from_parts: need parts >= 1, M >= 1, C >= 1");
    return hv_check_launch("hv_layernorm_from_parts");
}
int hv_groupnorm_from_parts(const hv_gn_parts_params* p, void* stream) {
    if (!p || !p->part1 || !p->scale || !p->shift || !p->gamma || !p->beta) return hv_fail(HV_EINVAL, "hv_groupnorm_from_parts: null");
    if (hvk_gn_from_parts(*p, (hipStream_t)stream) != 0)
        return hv_fail(HV_EINVAL, "hv_groupnorm_from_parts: need (C1 + C2) % groups == 0, parts >= 1, pixels >= 1");
    return hv_check_launch("hv_groupnorm_from_parts");
}

int hv_layernorm_stats(const uint16_t* X, long ldx, int M, int C, float eps, float* mean, float* rstd,
                       void* stream) {
    if (!X || !mean || !rstd || M <= 0) return hv_fail(HV_EINVAL, "hv_layernorm_stats: null");
    if (C % 8 != 0 || C > 2048) return hv_fail(HV_EINVAL, "hv_layernorm_stats: C % 8, C <= 2048");
    hvk_layernorm(X, ldx, M, C, eps, mean, rstd, (hipStream_t)stream);
    return hv_check_launch("hv_layernorm_stats");
}

int hv_attention(const hv_attention_params* p, void* stream) {
    if (!p || !p->Q || !p->K || !p->Vt || !p->O) return hv_fail(HV_EINVAL, "hv_attention: null operand");
    int rc = hvk_attention(*p, (hipStream_t)stream);
    if (rc == -2) return hv_fail(HV_ENOTSUP, "hv_attention: head dim must be 40, 80 or 160");
    if (rc != 0)
        return hv_fail(HV_EINVAL, "hv_attention: need 16-byte aligned strides, L1 % 8 == 0, L2 % 8 == 0, tensors below 4 GiB");
    return hv_check_launch("hv_attention");
}

int hv_attention_fp8_quantize(const uint16_t* K, long ldk, const uint16_t* Vt, long ldvt, int n_images, int heads, int D, int L,
                              float* kscale, float* vamax, const float* vfloor, uint8_t* K8, long ldk8, uint8_t* Vt8, long ldvt8,
                              int phase, void* stream) {
    if (!K || !Vt || !kscale || !vamax) return hv_fail(HV_EINVAL, "hv_attention_fp8_quantize: null operand");
    int rc = hvk_attention_fp8_quantize(K, ldk, Vt, ldvt, n_images, heads, D, L, kscale, vamax, vfloor, K8, ldk8, Vt8, ldvt8, phase,
                                        (hipStream_t)stream);
    if (rc == -2) return hv_fail(HV_ENOTSUP, "hv_attention_fp8_quantize: head dim must be 40, 80 or 160");
    if (rc != 0)
        return hv_fail(HV_EINVAL, "hv_attention_fp8_quantize: need L % 8 == 0, 8-byte aligned strides, phase 1..3, outputs for phase 2");
    return hv_check_launch("hv_attention_fp8_quantize");
}

int hv_attention_fp8(const hv_attention_params* p, const float* kscale, const float* vamax, const float* kscale2,
                     const float* vamax2, void* stream) {
    if (!p || !p->Q || !p->K || !p->Vt || !p->O || !kscale || !vamax) return hv_fail(HV_EINVAL, "hv_attention_fp8: null operand");
    int rc = hvk_attention_fp8(*p, kscale, vamax, kscale2, vamax2, (hipStream_t)stream);
    if (rc == -2) return hv_fail(HV_ENOTSUP, "hv_attention_fp8: head dim must be 40, 80 or 160");
    if (rc != 0)
        return hv_fail(HV_EINVAL, "hv_attention_fp8: transposed-V form only, L1 % 8 == 0, L2 % 8 == 0, 8-byte aligned strides, "
                                  "bank scales with the bank");
    return hv_check_launch("hv_attention_fp8");
}

static int g_hv_cmdlist_fallbacks = 0;  // command lists whose graph capture failed (hv_cmdlist_fallbacks())
static int g_hv_cmdlist_graphs = 1;  // hv_set_tuning(HV_TUNE_CMDLIST_GRAPHS): 0 = hv_cmdlist_run re-issues the closures on every run (A/B)
int hv_set_tuning(int key, int value) {
    if (key == HV_TUNE_ATTN_D40 && (value == 0 || value == 2)) hvk_attention_tune(40, value);
    else if (key == HV_TUNE_GEMM_MAX_GRID && value >= 8 && value % 8 == 0) hvk_gemm_tune(value);
    else if (key == HV_TUNE_GEMM_GLDS && ((value >= 0 && value <= 3) || value == 6)) hvk_gemm_use_glds(value);
    else if (key == HV_TUNE_GEMM_XS && (value == 0 || value == 1)) hvk_gemm_use_xs(value);
    else if (key == HV_TUNE_GEMM_WR && (value >= 0 && value <= 2)) hvk_gemm_use_wr(value);
    else if (key == HV_TUNE_GEMM_C4 && (value >= 0 && value <= 2)) hvk_gemm_use_c4(value);
    else if (key == HV_TUNE_GEMM_W4 && (value >= 0 && value <= 4)) hvk_gemm_use_w4(value);
    else if (key == HV_TUNE_CONV_GLDS && (value == 0 || value == 1)) hvk_conv_use_glds(value);
    else if (key == HV_TUNE_CONV_BIG && (value >= 0 && value <= 3)) hvk_conv_use_big(value);
    else if (key == HV_TUNE_CONV_RASTER && (value >= 0 && value <= 2)) hvk_conv_raster(value);
    else if (key == HV_TUNE_CONV_W4 && (value >= 0 && value <= 3)) hvk_conv_use_w4(value);
    else if (key == HV_TUNE_CMDLIST_GRAPHS && (value == 0 || value == 1)) g_hv_cmdlist_graphs = value;
    else return hv_fail(HV_EINVAL, "hv_set_tuning: unknown key/value");
    return HV_OK;
}

int hv_temporal_attention(const hv_temporal_attention_params* p, void* stream) {
    if (!p || !p->Q || !p->K || !p->V || !p->O) return hv_fail(HV_EINVAL, "hv_temporal_attention: null operand");
    int rc = hvk_temporal(*p, (hipStream_t)stream);
    if (rc == -2) return hv_fail(HV_ENOTSUP, "hv_temporal_attention: head dim must be 40, 80 or 160, F <= 32");
    if (rc != 0) return hv_fail(HV_EINVAL, "hv_temporal_attention: bad sizes");
    return hv_check_launch("hv_temporal_attention");
}

int hv_pack_ncfhw(const void* src, int src_is_bf16, int B, int C, int Fsrc, int H, int W, const int* frames, int F,
                  int rep, uint16_t* dst, int Cpad, void* stream) {
    if (!src || !dst || Cpad < C || Cpad % 8 != 0 || F <= 0) return hv_fail(HV_EINVAL, "hv_pack_ncfhw: bad args");
    hvk_pack(src, src_is_bf16, B, C, Fsrc, H, W, frames, F, rep, dst, Cpad, (hipStream_t)stream);
    return hv_check_launch("hv_pack_ncfhw");
}

int hv_unpack_nhwc(const uint16_t* src, int ldc, int B, int C, int F, int H, int W, void* dst, int dst_is_bf16,
                   void* stream) {
    if (!src || !dst) return hv_fail(HV_EINVAL, "hv_unpack_nhwc: null");
    hvk_unpack(src, ldc, B, C, F, H, W, dst, dst_is_bf16, (hipStream_t)stream);
    return hv_check_launch("hv_unpack_nhwc");
}

int hv_pixel_unshuffle(const float* src, int B, int C, int F, int H, int W, int r, uint16_t* dst, void* stream) {
    if (!src || !dst || H % r || W % r) return hv_fail(HV_EINVAL, "hv_pixel_unshuffle: bad args");
    hvk_unshuffle(src, B, C, F, H, W, r, dst, (hipStream_t)stream);
    return hv_check_launch("hv_pixel_unshuffle");
}

int hv_plucker_unshuffle(const float* K, const float* c2w, int F, int H, int W, int r, uint16_t* dst, void* stream) {
    if (!K || !c2w || !dst || F <= 0 || r <= 0 || H % r || W % r) return hv_fail(HV_EINVAL, "hv_plucker_unshuffle: bad args");
    hvk_plucker(K, c2w, F, H, W, r, dst, (hipStream_t)stream);
    return hv_check_launch("hv_plucker_unshuffle");
}

int hv_affine_apply(const uint16_t* X, long ldx, int rows, int rows_per_image, int C, const float* scale, const float* shift,
                    int act, uint16_t* Y, long ldy, void* stream) {
    if (!X || !Y || !scale || !shift) return hv_fail(HV_EINVAL, "hv_affine_apply: null operand");
    if (rows <= 0 || rows_per_image <= 0 || rows % rows_per_image != 0 || C <= 0 || C % 8 != 0 || C > 2560 || ldx % 8 != 0 || ldy % 8 != 0)
        return hv_fail(HV_EINVAL, "hv_affine_apply: need whole images (rows % rows_per_image == 0), C % 8 == 0, C <= 2560 and 16-byte aligned rows");
    hvk_affine_apply(X, ldx, rows, rows_per_image, C, nullptr, 0, 0, scale, shift, act, Y, ldy, (hipStream_t)stream);
    return hv_check_launch("hv_affine_apply");
}

int hv_affine_apply_cat(const uint16_t* X, long ldx, int C, const uint16_t* X2, long ldx2, int C2, int rows, int rows_per_image,
                        const float* scale, const float* shift, int act, uint16_t* Y, long ldy, void* stream) {
    if (!X || !X2 || !Y || !scale || !shift) return hv_fail(HV_EINVAL, "hv_affine_apply_cat: null operand");
    if (rows <= 0 || rows_per_image <= 0 || rows % rows_per_image != 0 || C <= 0 || C2 <= 0 || C % 8 != 0 || C2 % 8 != 0 || C + C2 > 2560 ||
        ldx % 8 != 0 || ldx2 % 8 != 0 || ldy % 8 != 0 || ldy < C + C2)
        return hv_fail(HV_EINVAL, "hv_affine_apply_cat: need whole images, C % 8 == 0, C2 % 8 == 0, C + C2 <= 2560, 16-byte aligned rows, ldy >= C + C2");
    hvk_affine_apply(X, ldx, rows, rows_per_image, C, X2, ldx2, C2, scale, shift, act, Y, ldy, (hipStream_t)stream);
    return hv_check_launch("hv_affine_apply_cat");
}

int hv_timestep_embedding(const float* t, int B, int dim, uint16_t* dst, void* stream) {
    if (!t || !dst || dim % 2) return hv_fail(HV_EINVAL, "hv_timestep_embedding: bad args");
    hvk_timestep(t, B, dim, dst, (hipStream_t)stream);
    return hv_check_launch("hv_timestep_embedding");
}

int hv_accumulate_window(const uint16_t* pred, int ldc, int rep, int C, int f_win, int H, int W, const int* frames,
                         int F, float* acc, float* counter, void* stream) {
    if (!pred || !frames || !acc || !counter) return hv_fail(HV_EINVAL, "hv_accumulate_window: null");
    hvk_accumulate(pred, ldc, rep, C, f_win, H, W, frames, F, acc, counter, (hipStream_t)stream);
    return hv_check_launch("hv_accumulate_window");
}

int hv_cfg_ddim_step(float* latents, float* acc, float* counter, int rep, int C, int F, int H, int W,
                     const float* coeffs, void* stream) {
    if (!latents || !acc || !counter || !coeffs || (rep != 1 && rep != 2))
        return hv_fail(HV_EINVAL, "hv_cfg_ddim_step: bad args");
    hvk_cfg_ddim(latents, acc, counter, rep, C, F, H, W, coeffs, (hipStream_t)stream);
    return hv_check_launch("hv_cfg_ddim_step");
}

int hv_cmdlist_begin(void) {
    if (g_hv_recording) return hv_fail(HV_EINVAL, "hv_cmdlist_begin: already recording");
    g_hv_recording = new HvCmdList();
    return HV_OK;
}
int hv_cmdlist_cut(void** list_out) {
    if (!g_hv_recording || !list_out) return hv_fail(HV_EINVAL, "hv_cmdlist_cut: not recording");
    *list_out = (void*)g_hv_recording;
    g_hv_recording = new HvCmdList();
    return HV_OK;
}
int hv_cmdlist_end(void** list_out) {
    if (!g_hv_recording || !list_out) return hv_fail(HV_EINVAL, "hv_cmdlist_end: not recording");
    *list_out = (void*)g_hv_recording;
    g_hv_recording = nullptr;
    return HV_OK;
}
int hv_cmdlist_size(void* list) { return list ? (int)((HvCmdList*)list)->cmds.size() : 0; }
int hv_cmdlist_run(void* list, void* stream) {
    if (!list) return hv_fail(HV_EINVAL, "hv_cmdlist_run: null list");
    HvCmdList* cl = (HvCmdList*)list;
#ifndef HV_EMU
    if (g_hv_cmdlist_graphs && !cl->cmds.empty() && !cl->no_graph) {
        if (cl->exec == nullptr) {
            // First run: capture the segment's launches into a graph instead of issuing them.  The capture stream is created
            // per capture on the CURRENT device and destroyed with it (a process-wide one is neither per-device nor
            // thread-safe, and a failed hipStreamEndCapture would leave it in capture mode for every later segment --
            // ADVICE round 4).  Any failure ends the capture, discards what was built and falls back to re-issuing the
            // closures for this list from now on: the launches of a failed capture were never executed.
            hipStream_t cs = nullptr;
            hipGraph_t g = nullptr;
            (void)hipGetLastError();  // a stale sticky error of an earlier, unrelated call must not fail this capture
            bool ok = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) == hipSuccess;
            if (ok) {
                ok = hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed) == hipSuccess;
                if (ok) {
                    for (auto& c : cl->cmds) c(cs);
                    const bool launched = hipGetLastError() == hipSuccess;  // a launch error inside the capture
                    ok = hipStreamEndCapture(cs, &g) == hipSuccess && g != nullptr && launched;
                }
                if (ok) ok = hipGraphInstantiate(&cl->exec, g, nullptr, nullptr, 0) == hipSuccess;
                if (g) (void)hipGraphDestroy(g);
                (void)hipStreamDestroy(cs);
            }
            if (!ok) {
                (void)hipGetLastError();  // clear the sticky error of the failed capture
                if (cl->exec) (void)hipGraphExecDestroy(cl->exec);
                cl->exec = nullptr;
                cl->no_graph = true;
                // not an error (the closures are re-issued below), but a silent slow path must be detectable: counted
                // (hv_cmdlist_fallbacks) and reported once per process
                static bool said = false;
                ++g_hv_cmdlist_fallbacks;
                if (!said) {
                    said = true;
                    fprintf(stderr, "humanvid_hip: graph capture of a command list failed; its %zu launches are re-issued one by one from now on\n",
                            cl->cmds.size());
                }
            }
        }
        if (cl->exec != nullptr) {
            const hipError_t e = hipGraphLaunch(cl->exec, (hipStream_t)stream);
            if (e != hipSuccess) return hv_fail(HV_EHIP, hipGetErrorString(e));
            return HV_OK;
        }
    }
#endif
    for (auto& c : cl->cmds) c((hipStream_t)stream);
    return hv_check_launch("hv_cmdlist_run");
}
int hv_cmdlist_fallbacks(void) { return g_hv_cmdlist_fallbacks; }
int hv_cmdlist_destroy(void* list) {
    delete (HvCmdList*)list;
    return HV_OK;
}

#ifndef HV_EMU
int hv_profile_begin(void) {
    if (g_hv_prof) return hv_fail(HV_EINVAL, "hv_profile_begin: a profile is already open");
    g_hv_prof = new HvProfile();
    return HV_OK;
}
// The reduced text of a profile whose first hv_profile_end() call came with too small a buffer: launches stop being
// recorded at that call (g_hv_prof is cleared), the text stays here until a call with enough room (or out == NULL with
// capacity < 0 to drop it) collects it -- the "call again with a larger buffer" contract of the header.
static thread_local std::string* g_hv_prof_text = nullptr;
int hv_profile_end(char* out, int capacity) {
    // -> lines "launches\ttotal_ms\tkey\n", one per distinct key, in first-seen order; returns the byte count needed
    if (!g_hv_prof && !g_hv_prof_text) return hv_fail(HV_EINVAL, "hv_profile_end: no open profile");
    if (g_hv_prof) {
        HvProfile* pr = g_hv_prof;
        g_hv_prof = nullptr;
        std::vector<std::string> keys;
        std::vector<double> ms;
        std::vector<int> cnt;
        std::string seq;  // launch order as indices into `keys` (tools/pmc_by_shape.py lines rocprofv3 dispatches up with it)
        for (auto& en : pr->entries) {
            float t = 0.f;
            (void)hipEventSynchronize(en.e1);
            (void)hipEventElapsedTime(&t, en.e0, en.e1);
            (void)hipEventDestroy(en.e0);
            (void)hipEventDestroy(en.e1);
            size_t i = 0;
            for (; i < keys.size(); ++i)
                if (keys[i] == en.key) break;
            if (i == keys.size()) {
                keys.push_back(en.key);
                ms.push_back(0.0);
                cnt.push_back(0);
            }
            ms[i] += t;
            cnt[i] += 1;
            seq += (seq.empty() ? "" : ",") + std::to_string(i);
        }
        delete pr;
        delete g_hv_prof_text;
        g_hv_prof_text = new std::string();
        char line[320];
        for (size_t i = 0; i < keys.size(); ++i) {
            snprintf(line, sizeof(line), "%d\t%.6f\t%s\n", cnt[i], ms[i], keys[i].c_str());
            *g_hv_prof_text += line;
        }
        *g_hv_prof_text += "#seq\t" + seq + "\n";
    }
    const int need = (int)g_hv_prof_text->size() + 1;
    if (out && capacity >= need) {
        memcpy(out, g_hv_prof_text->c_str(), (size_t)need);
    } else if (!(out == nullptr && capacity < 0)) {
        if (out && capacity > 0) out[0] = 0;
        return need;  // too small (or a size query): the text is kept for the next call
    }
    delete g_hv_prof_text;
    g_hv_prof_text = nullptr;
    return need;
}
int hv_graph_begin(void* stream) {
    hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) return hv_fail(HV_EHIP, hipGetErrorString(e));
    return HV_OK;
}
int hv_graph_end(void* stream, void** graph_exec_out) {
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
    if (e != hipSuccess) return hv_fail(HV_EHIP, hipGetErrorString(e));
    hipGraphExec_t ex = nullptr;
    e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return hv_fail(HV_EHIP, hipGetErrorString(e));
    *graph_exec_out = (void*)ex;
    return HV_OK;
}
int hv_graph_launch(void* graph_exec, void* stream) {
    hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
    if (e != hipSuccess) return hv_fail(HV_EHIP, hipGetErrorString(e));
    return HV_OK;
}
int hv_graph_destroy(void* graph_exec) {
    (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
    return HV_OK;
}
int hv_event_create(void** ev) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return hv_fail(HV_EHIP, "hipEventCreate");
    *ev = (void*)e;
    return HV_OK;
}
int hv_event_record(void* ev, void* stream) {
    if (hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) != hipSuccess) return hv_fail(HV_EHIP, "hipEventRecord");
    return HV_OK;
}
int hv_event_elapsed_ms(void* start, void* stop, float* ms) {
    if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess) return hv_fail(HV_EHIP, "hipEventSynchronize");
    if (hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess)
        return hv_fail(HV_EHIP, "hipEventElapsedTime");
    return HV_OK;
}
int hv_event_destroy(void* ev) {
    (void)hipEventDestroy((hipEvent_t)ev);
    return HV_OK;
}
#else
int hv_profile_begin(void) { return hv_fail(HV_ENOTSUP, "emulator build"); }
int hv_profile_end(char*, int) { return hv_fail(HV_ENOTSUP, "emulator build"); }
int hv_graph_begin(void*) { return hv_fail(HV_ENOTSUP, "emulator build"); }
int hv_graph_end(void*, void**) { return hv_fail(HV_ENOTSUP, "emulator build"); }
int hv_graph_launch(void*, void*) { return hv_fail(HV_ENOTSUP, "emulator build"); }
int hv_graph_destroy(void*) { return HV_OK; }
int hv_event_create(void**) { return hv_fail(HV_ENOTSUP, "emulator build"); }
int hv_event_record(void*, void*) { return hv_fail(HV_ENOTSUP, "emulator build"); }
int hv_event_elapsed_ms(void*, void*, float*) { return hv_fail(HV_ENOTSUP, "emulator build"); }
int hv_event_destroy(void*) { return HV_OK; }
#endif

}  // extern "C"
