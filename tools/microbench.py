#!/usr/bin/env python3
"""Per-kernel timing at the config-#3 shapes (24f x 768x512 -> n = 48 images, 96x64 latents).

Prints one line per kernel: average milliseconds (hip events on the launch stream), algorithmic
TFLOP/s and GB/s, and writes gpurun_out/microbench.json.  Synthetic random bf16 data.
Usage: python tools/microbench.py [--quick] [--only gemm,conv,attn,temporal,norm]
"""
import argparse
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from humanvid_amd import _abi as A
from humanvid_amd import lib as hvlib
from humanvid_amd import ops

BF16 = torch.bfloat16


def timeit(fn, iters=5, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def c4_trace(L):
    """timing build of hv_conv_w4_kernel (-DHV_C4_TRACE): s_memtime marks of the last launch, per workgroup"""
    if not hasattr(L.cdll, "hv_c4_trace_read"):
        return
    import ctypes
    import numpy as np
    import torch
    buf = np.zeros(2048 * 8, dtype=np.uint64)
    torch.cuda.synchronize()
    L.cdll.hv_c4_trace_read(buf.ctypes.data_as(ctypes.c_void_p))
    t = buf.reshape(2048, 8).astype(np.float64)
    t = t[t[:, 4] > t[:, 0]]
    if not len(t):
        return
    t0 = t[:, 0].min()
    d = np.diff(t[:, :5], axis=1)
    print(f"    c4 trace ({len(t)} workgroups, s_memtime ticks): prologue {d[:, 0].mean():.0f}, k-loop {d[:, 1].mean():.0f}, epilogue load issue "
          f"{d[:, 2].mean():.0f}, arithmetic + stores {d[:, 3].mean():.0f}; launch span {t[:, 4].max() - t0:.0f}; workgroup starts (sorted, every 128th): "
          + " ".join(f"{v - t0:.0f}" for v in np.sort(t[:, 0])[::128]), flush=True)
    buf[:] = 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--images", type=int, default=48)
    args = ap.parse_args()
    only = set(args.only.split(",")) if args.only else None
    dev = hvlib.require_gpu()
    L = A.HvLibrary(os.environ["HV_LIB"]) if os.environ.get("HV_LIB") else hvlib.load()  # A/B of build variants
    st = hvlib.current_stream()
    if os.environ.get("HV_GEMM_GLDS"):
        L.call("hv_set_tuning", 3, int(os.environ["HV_GEMM_GLDS"]))  # A/B of the GEMM kernel variants
    for kv in filter(None, os.environ.get("HV_TUNE", "").split(",")):  # any tuning key of the loaded library: HV_TUNE="8=1,3=2"
        L.call("hv_set_tuning", int(kv.split("=")[0]), int(kv.split("=")[1]))
    if os.environ.get("HV_CONV_BIG"):
        L.call("hv_set_tuning", 5, int(os.environ["HV_CONV_BIG"]))
    if os.environ.get("HV_CONV_GLDS"):
        L.call("hv_set_tuning", 4, int(os.environ["HV_CONV_GLDS"]))
    n = args.images
    out = []

    def rnd(*shape, dtype=BF16, scale=1.0):
        return (torch.randn(*shape, device=dev) * scale).to(dtype)

    def report(name, ms, flops, bytes_):
        rec = dict(kernel=name, ms=ms, tflops=flops / ms / 1e9, gbs=bytes_ / ms / 1e6)
        out.append(rec)
        print(f"{name:58s} {ms:9.3f} ms  {rec['tflops']:8.1f} TF/s  {rec['gbs']:8.1f} GB/s", flush=True)

    levels = [(96 * 64, 320, 96, 64), (48 * 32, 640, 48, 32), (24 * 16, 1280, 24, 16), (12 * 8, 1280, 12, 8)]

    if only is None or "gemm" in only:
        only_l0 = os.environ.get("HV_MB_ONLY_L0") == "1"  # counter passes: the two streamed level-0 projections only
        for (N_tok, C, _, _) in levels[:1] if only_l0 else levels[:3]:
            M = n * N_tok
            x = rnd(M, C)
            for name, Nn, K, geglu in [("qkv", 3 * C, C, False), ("proj", C, C, False), ("ff1_geglu", 8 * C, C, True),
                                       ("ff2", C, 4 * C, False)]:
                if only_l0 and name not in ("qkv", "ff1_geglu"):
                    continue
                w = rnd(Nn, K, scale=K**-0.5)
                xx = x if K == C else rnd(M, K)
                y = torch.empty(M, Nn // 2 if geglu else Nn, dtype=BF16, device=dev)
                bias = torch.zeros(Nn, device=dev)
                # the epilogue forms of the engine: LayerNorm fold on the QKV / feed-forward input projections,
                # in-place residual on the output projections
                kw = {}
                if name in ("qkv", "ff1_geglu"):
                    kw = dict(row_mean=torch.zeros(M, device=dev), row_rstd=torch.ones(M, device=dev),
                              colsum=torch.zeros(Nn, device=dev))
                else:
                    kw = dict(residual=y)
                ms = timeit(lambda: ops.gemm(L, st, xx, w, y, bias=bias, geglu=geglu, **kw))
                report(f"gemm {name} M={M} N={Nn} K={K}", ms, 2.0 * M * Nn * K, 2.0 * (M * K + Nn * K + y.numel()))
                if hasattr(L.cdll, "hv_w4_trace_read"):  # timing build of hv_gemm_w4_kernel (-DHV_W4_TRACE): where a workgroup's time goes
                    import ctypes
                    import numpy as np
                    buf = np.zeros(256 * 8, dtype=np.uint64)
                    torch.cuda.synchronize()
                    L.cdll.hv_w4_trace_read(buf.ctypes.data_as(ctypes.c_void_p))
                    t = buf.reshape(256, 8).astype(np.float64)
                    t = t[t[:, 4] > 0]
                    if len(t):
                        m = t.mean(0)
                        print(f"    w4 trace (s_memtime ticks, mean over {len(t)} workgroups): kernel {m[0]:.0f}, vmcnt waits {m[1]:.0f}, barriers {m[2]:.0f}, "
                              f"epilogue {m[3]:.0f}; {m[4]:.0f} k-tiles, {m[5]:.0f} epilogues -> per k-tile: wait {m[1] / m[4]:.1f}, barrier {m[2] / m[4]:.1f}, "
                              f"rest {(m[0] - m[1] - m[2] - m[3]) / m[4]:.1f}; per epilogue {m[3] / max(m[5], 1):.1f}", flush=True)
                del w, y
        if only_l0:
            return
        # fused-prologue / LN-fold variants at level 0
        M, C = n * 6144, 320
        x, w = rnd(M, C), rnd(C, C, scale=C**-0.5)
        y = torch.empty(M, C, dtype=BF16, device=dev)
        sc, sh = torch.ones(n, C, device=dev), torch.zeros(n, C, device=dev)
        ms = timeit(lambda: ops.gemm(L, st, x, w, y, pro_scale=sc, pro_shift=sh, rows_per_image=6144))
        report("gemm proj_in + GN prologue (level 0)", ms, 2.0 * M * C * C, 4.0 * M * C)
        mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
        cs = torch.zeros(C, device=dev)
        ms = timeit(lambda: ops.gemm(L, st, x, w, y, row_mean=mean, row_rstd=rstd, colsum=cs, residual=x))
        report("gemm + LN fold + residual (level 0)", ms, 2.0 * M * C * C, 6.0 * M * C)
        ms = timeit(lambda: ops.layernorm_stats(L, st, x, mean, rstd))
        report("layernorm stats (level 0)", ms, 0.0, 2.0 * M * C)
        # the level-0 output projections as the engine launches them: residual in place + LayerNorm / GroupNorm parts of the output
        yres = rnd(M, C)
        bias0 = torch.zeros(C, device=dev)
        npl = ops.gemm(L, st, x, w, yres, bias=bias0, residual=yres, query_ln_parts=True)
        if npl > 0:
            lpart = torch.zeros(M, npl, 2, device=dev)
            ms = timeit(lambda: ops.gemm(L, st, x, w, yres, bias=bias0, residual=yres, ln_part=lpart))
            report(f"gemm out-proj + residual + ln parts ({npl}) (level 0)", ms, 2.0 * M * C * C, 6.0 * M * C)
        npg = ops.gemm(L, st, x, w, yres, bias=bias0, residual=yres, gn_rows_per_image=6144, query_gn_parts=True)
        if npg > 0:
            gpart = torch.zeros(n, npg, C, 2, device=dev)
            ms = timeit(lambda: ops.gemm(L, st, x, w, yres, bias=bias0, residual=yres, gn_part=gpart, gn_rows_per_image=6144))
            report(f"gemm out-proj + residual + gn parts ({npg}) (level 0)", ms, 2.0 * M * C * C, 6.0 * M * C)

    if only is None or "conv" in only:
        for (_, C, H, W) in levels:
            x = rnd(n, H, W, C)
            w = rnd(C, 9, C, scale=(9 * C) ** -0.5)
            y = torch.empty(n, H, W, C, dtype=BF16, device=dev)
            sc, sh = torch.ones(n, C, device=dev), torch.zeros(n, C, device=dev)
            bias = torch.zeros(C, device=dev)
            ms = timeit(lambda: ops.conv3x3(L, st, x, w, y, pro_scale=sc, pro_shift=sh, pro_act=A.ACT_SILU, bias=bias))
            report(f"conv3x3 GN+SiLU fused {C}->{C} @{H}x{W}", ms, 2.0 * n * H * W * C * C * 9, 4.0 * n * H * W * C)
            ms = timeit(lambda: ops.conv3x3(L, st, x, w, y, bias=bias))
            report(f"conv3x3 plain {C}->{C} @{H}x{W}", ms, 2.0 * n * H * W * C * C * 9, 4.0 * n * H * W * C)
            c4_trace(L)
        # the up path's plain convolutions (the concatenated, normalised activation comes from hv_affine_apply_cat)
        for (Ci, Co, H, W) in ((640, 320, 96, 64), (960, 320, 96, 64), (1280, 640, 48, 32), (1920, 640, 48, 32), (2560, 1280, 24, 16)):
            x = rnd(n, H, W, Ci)
            w = rnd(Co, 9, Ci, scale=(9 * Ci) ** -0.5)
            y = torch.empty(n, H, W, Co, dtype=BF16, device=dev)
            res = rnd(n, H, W, Co)
            ms = timeit(lambda: ops.conv3x3(L, st, x, w, y, residual=res))
            report(f"conv3x3 plain {Ci}->{Co} @{H}x{W} + residual", ms, 2.0 * n * H * W * Ci * Co * 9, 2.0 * n * H * W * (Ci + 2 * Co))
            c4_trace(L)
            del x, w, y, res
        # two-source + upsample + stride 2 at representative sizes
        x1, x2 = rnd(n, 48, 32, 640), rnd(n, 48, 32, 320)
        w = rnd(640, 9, 960, scale=0.01)
        y = torch.empty(n, 48, 32, 640, dtype=BF16, device=dev)
        ms = timeit(lambda: ops.conv3x3(L, st, x1, w, y, x2=x2))
        report("conv3x3 two-source 640+320->640 @48x32", ms, 2.0 * n * 48 * 32 * 960 * 640 * 9, 2.0 * n * 48 * 32 * 1600)
        x = rnd(n, 48, 32, 640)
        w = rnd(640, 9, 640, scale=0.01)
        y = torch.empty(n, 96, 64, 640, dtype=BF16, device=dev)
        ms = timeit(lambda: ops.conv3x3(L, st, x, w, y, mode=A.CONV_UP2))
        report("conv3x3 upsample-folded 640 @48x32->96x64", ms, 2.0 * n * 96 * 64 * 640 * 640 * 9, 2.0 * n * 640 * (48 * 32 + 96 * 64))
        c4_trace(L)
        for (C, H, W) in ((1280, 24, 16), (1280, 12, 8)):
            x = rnd(n, H, W, C)
            w = rnd(C, 9, C, scale=0.01)
            y = torch.empty(n, 2 * H, 2 * W, C, dtype=BF16, device=dev)
            ms = timeit(lambda: ops.conv3x3(L, st, x, w, y, mode=A.CONV_UP2))
            report(f"conv3x3 upsample-folded {C} @{H}x{W}->{2 * H}x{2 * W}", ms, 2.0 * n * 4 * H * W * C * C * 9, 2.0 * n * C * 5 * H * W)
            del x, w, y
        x = rnd(n, 96, 64, 320)
        w = rnd(320, 9, 320, scale=0.01)
        y = torch.empty(n, 48, 32, 320, dtype=BF16, device=dev)
        ms = timeit(lambda: ops.conv3x3(L, st, x, w, y, mode=A.CONV_S2))
        report("conv3x3 stride-2 320 @96x64->48x32", ms, 2.0 * n * 48 * 32 * 320 * 320 * 9, 2.0 * n * 320 * (48 * 32 + 96 * 64))

    if only is None or "norm" in only:
        for (_, C, H, W) in levels[:3]:
            x = rnd(n, H, W, C)
            g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
            part = torch.zeros(n * 64 * 32 * 2, device=dev)
            sc, sh = torch.zeros(n, C, device=dev), torch.zeros(n, C, device=dev)
            ms = timeit(lambda: ops.groupnorm_affine(L, st, x, g, b, 32, 1e-5, part, sc, sh))
            report(f"groupnorm stats C={C} @{H}x{W}", ms, 0.0, 2.0 * x.numel())

    if only is None or "attn" in only:
        for (N_tok, C, _, _) in levels:
            D = C // 8
            M = n * N_tok
            qk = rnd(M, 2 * C)
            vt = rnd(C, M)
            k2, vt2 = rnd(2 * N_tok, C), rnd(C, 2 * N_tok)
            o = torch.empty(M, C, dtype=BF16, device=dev)
            sel = torch.tensor([-1] * (n // 2) + [1] * (n - n // 2), dtype=torch.int32, device=dev)
            variants = [None]
            if D == 40:
                variants = [2, 0]  # generic kernel, hv_attention40 (default)
            for var in variants:
                if var is not None:
                    L.call("hv_set_tuning", 0, var)  # (key 0: the head-dim-40 kernel choice -- the only attention tuning key)
                ms = timeit(lambda: ops.attention(L, st, qk, qk[:, C:], vt, o, n_images=n, heads=8, D=D, Lq=N_tok,
                                                  L1=N_tok, ldq=2 * C, ldk=2 * C, ldvt=M, ldo=C, k2=k2, vt2=vt2,
                                                  ldk2=C, ldvt2=2 * N_tok, L2=N_tok, bank_sel=sel),
                            iters=3, warmup=1)
                flops = 4.0 * 8 * D * N_tok * (N_tok * (n // 2) + 2 * N_tok * (n - n // 2))
                report(f"attention D={D} N={N_tok} (cond half +bank) QT={var}", ms, flops, 2.0 * 4 * M * C)

    if only is None or "temporal" in only:
        for (N_tok, C, _, _) in levels:
            D = C // 8
            M = n * N_tok
            qkv = rnd(M, 3 * C)
            o = torch.empty(M, C, dtype=BF16, device=dev)
            ms = timeit(lambda: ops.temporal_attention(L, st, qkv, o, B=2, F=n // 2, P=N_tok, heads=8, D=D))
            Fr = n // 2
            report(f"temporal attention D={D} P={N_tok} F={Fr}", ms, 4.0 * 2 * N_tok * 8 * Fr * Fr * D, 2.0 * 4 * M * C)

    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/microbench.json", "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
