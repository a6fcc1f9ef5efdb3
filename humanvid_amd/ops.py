"""Typed Python entry points over the C ABI: build the parameter structs from tensors.

Tensors are only used for their `data_ptr()`, shape and stride -- PyTorch is the allocator, not
the compute engine.  Every function takes the loaded library (`HvLibrary`) and a raw stream handle.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch

from . import _abi as A

BF16 = torch.bfloat16


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, dtype, what: str):
    if t.dtype != dtype:
        raise ValueError(f"{what}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{what}: must be contiguous")


def gemm(lib, stream, x, w, y, *, x2=None, k1=0, yt=None, n_split=0, pro_scale=None, pro_shift=None,
         rows_per_image=1, pro_act=A.ACT_NONE, bias=None, row_mean=None, row_rstd=None, colsum=None, pe=None,
         pe_period=1, pe_frames=1, rowvec=None, rowvec_period=1, residual=None, geglu=False, out_act=A.ACT_NONE,
         M=None, ldx=None, ldy=None, ldr=None, ldx2=None, row_perm=None, gn_part=None, gn_rows_per_image=0,
         query_gn_parts=False, ln_part=None, query_ln_parts=False):
    """y[M, N(/2)] = epi(pro(x)[M, K] @ w[N, K]^T); x/x2/y/residual may be row-strided views.
    row_perm=(X, Y, P): output (and residual) row (x*Y + y)*P + p is taken at (y*X + x)*P + p.
    gn_part [M / gn_rows_per_image, parts, N, 2] fp32: GroupNorm partial statistics of the output (hv_gemm_gn_parts);
    query_gn_parts=True only asks how many parts per image this problem would write (0 = cannot) and launches nothing.
    ln_part [M, parts, 2] fp32 (parts = what query_ln_parts returns: N / 64, or 4 from hv_gemm_wr_kernel) / query_ln_parts: the same for the LayerNorm row statistics of the output."""
    N, K = w.shape
    M = x.shape[0] if M is None else M
    p = A.GemmParams(
        X=_p(x), ldx=x.stride(0) if ldx is None else ldx, X2=_p(x2),
        ldx2=(x2.stride(0) if x2 is not None else 0) if ldx2 is None else ldx2, K1=k1,
        W=_p(w), Y=_p(y), ldy=y.stride(0) if ldy is None else ldy, out_f32=int(y.dtype == torch.float32),
        Yt=_p(yt), ldyt=yt.stride(0) if yt is not None else 0, n_split=n_split,
        M=M, N=N, K=K, pro_scale=_p(pro_scale), pro_shift=_p(pro_shift), rows_per_image=rows_per_image,
        pro_act=pro_act, bias=_p(bias), row_mean=_p(row_mean), row_rstd=_p(row_rstd), colsum=_p(colsum),
        pe=_p(pe), pe_period=pe_period, pe_frames=pe_frames, rowvec=_p(rowvec), rowvec_period=rowvec_period,
        residual=_p(residual), ldr=(residual.stride(0) if residual is not None else 0) if ldr is None else ldr,
        geglu=int(geglu), out_act=out_act,
        perm_x=row_perm[0] if row_perm else 0, perm_y=row_perm[1] if row_perm else 0,
        perm_p=row_perm[2] if row_perm else 0,
        gn_part=_p(gn_part), gn_rows_per_image=gn_rows_per_image, ln_part=_p(ln_part),
    )
    if query_gn_parts:
        return int(lib.cdll.hv_gemm_gn_parts(C.byref(p)))
    if query_ln_parts:
        return int(lib.cdll.hv_gemm_ln_parts(C.byref(p)))
    lib.call("hv_gemm", C.byref(p), stream)


def affine_apply(lib, stream, x, scale, shift, y, *, rows_per_image, act=A.ACT_NONE, x2=None):
    """y[row, c] = act(x[row, c] * scale[row // rows_per_image, c] + shift[...]) -- GroupNorm apply as its own pass
    (x, y: [rows, C] bf16, possibly row-strided; scale / shift [images, C] fp32 from groupnorm_affine).
    x2 [rows, C2]: the channel concatenation [x | x2] is normalised into y [rows, C + C2] (scale / shift [images, C + C2])."""
    rows, Cc = x.shape
    if x2 is None:
        lib.call("hv_affine_apply", _p(x), x.stride(0), rows, rows_per_image, Cc, _p(scale), _p(shift), act, _p(y),
                 y.stride(0), stream)
        return
    assert x2.shape[0] == rows and y.shape[1] == Cc + x2.shape[1] and scale.shape[1] == y.shape[1]
    lib.call("hv_affine_apply_cat", _p(x), x.stride(0), Cc, _p(x2), x2.stride(0), x2.shape[1], rows, rows_per_image, _p(scale),
             _p(shift), act, _p(y), y.stride(0), stream)


def conv3x3(lib, stream, x, w, y, *, x2=None, mode=A.CONV_S1, pro_scale=None, pro_shift=None, pro_act=A.ACT_NONE,
            bias=None, rowvec=None, images_per_rowvec=1, rowvec_ld=None, residual=None, out_act=A.ACT_NONE,
            gn_part=None, query_gn_parts=False):
    """x [n,Hs,Ws,C1] (+x2 [n,Hs,Ws,C2]); w packed [Cout, 9, C1+C2]; y [n,Ho,Wo,Cout].
    gn_part [n, parts, Cout, 2] fp32: GroupNorm partial statistics of the output; query_gn_parts=True only returns `parts`."""
    n, Hs, Ws, C1 = x.shape
    _, Ho, Wo, Cout = y.shape
    p = A.Conv3x3Params(
        X=_p(x), C1=C1, X2=_p(x2), C2=0 if x2 is None else x2.shape[3], W=_p(w), Y=_p(y),
        n_images=n, Hs=Hs, Ws=Ws, Ho=Ho, Wo=Wo, Cout=Cout, mode=mode,
        pro_scale=_p(pro_scale), pro_shift=_p(pro_shift), pro_act=pro_act, bias=_p(bias),
        rowvec=_p(rowvec), images_per_rowvec=images_per_rowvec,
        rowvec_ld=(Cout if rowvec_ld is None else rowvec_ld), residual=_p(residual),
        residual_images=0 if residual is None else residual.shape[0], out_act=out_act, gn_part=_p(gn_part),
    )
    if query_gn_parts:
        return int(lib.cdll.hv_conv3x3_gn_parts(C.byref(p)))
    lib.call("hv_conv3x3", C.byref(p), stream)


def groupnorm_affine(lib, stream, x, gamma, beta, groups, eps, partial, scale, shift, *, x2=None, splits=None):
    n, pixels, C1 = x.shape[0], x.shape[1] * x.shape[2], x.shape[3]
    if splits is None:
        splits = max(1, min(64, pixels // 64))
    assert partial.numel() >= n * splits * groups * 2
    p = A.GroupNormParams(
        X=_p(x), C1=C1, X2=_p(x2), C2=0 if x2 is None else x2.shape[3], n_images=n, pixels=pixels,
        groups=groups, eps=eps, gamma=_p(gamma), beta=_p(beta), partial=_p(partial), splits=splits,
        scale=_p(scale), shift=_p(shift),
    )
    lib.call("hv_groupnorm_affine", C.byref(p), stream)


def groupnorm_from_parts(lib, stream, part1, gamma, beta, groups, eps, pixels, scale, shift, *, part2=None):
    """scale / shift [n, C1 + C2] from the partial statistics the producing kernels left: part [n, parts, C, 2] fp32."""
    n, parts1, C1, _ = part1.shape
    p = A.GnPartsParams(
        part1=_p(part1), parts1=parts1, C1=C1, part2=_p(part2), parts2=0 if part2 is None else part2.shape[1],
        C2=0 if part2 is None else part2.shape[2], n_images=n, pixels=pixels, groups=groups, eps=eps,
        gamma=_p(gamma), beta=_p(beta), scale=_p(scale), shift=_p(shift),
    )
    lib.call("hv_groupnorm_from_parts", C.byref(p), stream)


def layernorm_from_parts(lib, stream, part, C, mean, rstd, eps=1e-5):
    """mean / rstd [M] from the partial row sums the producing GEMM left: part [M, C / 64, 2] fp32"""
    M, parts, _ = part.shape
    lib.call("hv_layernorm_from_parts", part.data_ptr(), parts, M, C, eps, mean.data_ptr(), rstd.data_ptr(), stream)


def layernorm_stats(lib, stream, x, mean, rstd, eps=1e-5, M=None):
    M = x.shape[0] if M is None else M
    lib.call("hv_layernorm_stats", x.data_ptr(), x.stride(0), M, x.shape[1], eps, mean.data_ptr(), rstd.data_ptr(),
             stream)


def attention(lib, stream, q, k, vt, o, *, n_images, heads, D, Lq, L1, ldq, ldk, ldvt, ldo, k2=None, vt2=None,
              ldk2=0, ldvt2=0, L2=0, bank_sel=None):
    """vt / vt2: TRANSPOSED values [C][tokens] (the QKV GEMM epilogue writes them so)."""
    p = A.AttentionParams(
        Q=_p(q), ldq=ldq, K=_p(k), ldk=ldk, Vt=_p(vt), ldvt=ldvt, K2=_p(k2), ldk2=ldk2, Vt2=_p(vt2), ldvt2=ldvt2,
        bank_sel=_p(bank_sel), O=_p(o), ldo=ldo, n_images=n_images, heads=heads, D=D, Lq=Lq, L1=L1, L2=L2,
        scale=1.0 / math.sqrt(D),
    )
    lib.call("hv_attention", C.byref(p), stream)


def attention_fp8_quantize(lib, stream, k, vt, kscale, vamax, *, n_images, heads, D, L, ldk, ldvt, phase, vfloor=None,
                           k8=None, vt8=None):
    """fp8 pre-pass over one key source.  phase 1: kscale [n_images, heads, ceil(L/64)], vamax [heads] (fp32);
    phase 2: k8 [n_images*L, C] / vt8 [C, n_images*L] uint8 (e4m3), V scale = max(vamax, vfloor) / 384; 3: both."""
    lib.call("hv_attention_fp8_quantize", _p(k), ldk, _p(vt), ldvt, n_images, heads, D, L, _p(kscale), _p(vamax), _p(vfloor),
             _p(k8), k8.stride(0) if k8 is not None else 0, _p(vt8), vt8.stride(0) if vt8 is not None else 0, phase, stream)


def attention_fp8(lib, stream, q, k, vt, o, kscale, vamax, *, n_images, heads, D, Lq, L1, ldq, ldk, ldvt, ldo, k2=None,
                  vt2=None, ldk2=0, ldvt2=0, L2=0, bank_sel=None, kscale2=None, vamax2=None):
    """hv_attention on the fp8 MFMA (transposed-V form): k / vt / k2 / vt2 are the e4m3 (uint8) tensors and scales that
    attention_fp8_quantize produced for the own keys and the bank; q is bf16."""
    p = A.AttentionParams(
        Q=_p(q), ldq=ldq, K=_p(k), ldk=ldk, Vt=_p(vt), ldvt=ldvt, K2=_p(k2), ldk2=ldk2, Vt2=_p(vt2), ldvt2=ldvt2,
        bank_sel=_p(bank_sel), O=_p(o), ldo=ldo, n_images=n_images, heads=heads, D=D, Lq=Lq, L1=L1, L2=L2,
        scale=1.0 / math.sqrt(D),
    )
    lib.call("hv_attention_fp8", C.byref(p), _p(kscale), _p(vamax), _p(kscale2), _p(vamax2), stream)


def temporal_attention(lib, stream, qkv, o, *, B, F, P, heads, D):
    """Single-device form: qkv [(B F P), 3C] rows (b*F + f)*P + p with columns [q | k | v]."""
    Cc = heads * D
    ld = qkv.stride(0)
    eb = qkv.element_size()
    p = A.TemporalAttentionParams(
        Q=qkv.data_ptr(), ldq=ld, K=qkv.data_ptr() + Cc * eb, V=qkv.data_ptr() + 2 * Cc * eb, ldkv=ld,
        kv_stride_b=F * P, kv_stride_chunk=0, kv_chunk=F, O=_p(o), ldo=o.stride(0),
        B=B, Fq=F, Fkv=F, P=P, heads=heads, D=D, scale=1.0 / math.sqrt(D),
    )
    lib.call("hv_temporal_attention", C.byref(p), stream)


def temporal_attention_exchanged(lib, stream, qkv_recv, o_send, *, B, F_local, ranks, P, heads, D):
    """All-to-all form: qkv_recv [ranks, B, F_local, P, 3C] is what the frames -> pixels all-to-all delivered (chunk s = the
    frames of rank s, for this rank's P pixels); attends over all ranks * F_local frames and writes o_send
    [ranks, B, F_local, P, C] in the same order, which is exactly the send buffer of the returning all-to-all."""
    Cc = heads * D
    eb = qkv_recv.element_size()
    ld = 3 * Cc
    p = A.TemporalAttentionParams(
        Q=qkv_recv.data_ptr(), ldq=ld, K=qkv_recv.data_ptr() + Cc * eb, V=qkv_recv.data_ptr() + 2 * Cc * eb, ldkv=ld,
        kv_stride_b=F_local * P, kv_stride_chunk=B * F_local * P, kv_chunk=F_local, O=o_send.data_ptr(), ldo=Cc,
        B=B, Fq=F_local * ranks, Fkv=F_local * ranks, P=P, heads=heads, D=D, scale=1.0 / math.sqrt(D), qo_chunked=1,
    )
    lib.call("hv_temporal_attention", C.byref(p), stream)


def temporal_attention_sharded(lib, stream, q, kv_gathered, o, *, B, Fq, ranks, P, heads, D):
    """Frame-sharded form: q [(B Fq P), C] local query frames; kv_gathered [ranks, B, Fq, P, 2C]
    (all-gathered [k | v] of every rank's frames); o like q."""
    Cc = heads * D
    eb = q.element_size()
    p = A.TemporalAttentionParams(
        Q=q.data_ptr(), ldq=q.stride(0), K=kv_gathered.data_ptr(), V=kv_gathered.data_ptr() + Cc * eb, ldkv=2 * Cc,
        kv_stride_b=Fq * P, kv_stride_chunk=B * Fq * P, kv_chunk=Fq, O=_p(o), ldo=o.stride(0),
        B=B, Fq=Fq, Fkv=Fq * ranks, P=P, heads=heads, D=D, scale=1.0 / math.sqrt(D),
    )
    lib.call("hv_temporal_attention", C.byref(p), stream)


def pack_ncfhw(lib, stream, src, dst, rep=1, frames=None):
    """src [B,C,Fsrc,H,W] fp32|bf16 -> dst [(rep B) F, H, W, Cpad] bf16; `frames` (device int32 [F])
    selects / reorders source frames (context window, frame shard)."""
    B, Cc, Fs, H, W = src.shape
    Fr = Fs if frames is None else frames.numel()
    lib.call("hv_pack_ncfhw", src.data_ptr(), int(src.dtype == BF16), B, Cc, Fs, H, W, _p(frames), Fr, rep,
             dst.data_ptr(), dst.shape[-1], stream)


def unpack_nhwc(lib, stream, src, dst):
    """src [(B F), H, W, ldc] bf16 -> dst [B,C,F,H,W] fp32|bf16."""
    B, Cc, Fr, H, W = dst.shape
    lib.call("hv_unpack_nhwc", src.data_ptr(), src.shape[-1], B, Cc, Fr, H, W, dst.data_ptr(),
             int(dst.dtype == BF16), stream)


def pixel_unshuffle(lib, stream, src, dst, r):
    B, Cc, Fr, H, W = src.shape
    lib.call("hv_pixel_unshuffle", src.data_ptr(), B, Cc, Fr, H, W, r, dst.data_ptr(), stream)


def plucker_unshuffle(lib, stream, K, c2w, H, W, r, dst):
    """K [F,4] fp32 (fx, fy, cx, cy in pixels), c2w [F,4,4] fp32 -> dst [F, H/r, W/r, 6*r*r] bf16."""
    lib.call("hv_plucker_unshuffle", K.data_ptr(), c2w.data_ptr(), K.shape[0], H, W, r, dst.data_ptr(), stream)


def timestep_embedding(lib, stream, t, dst):
    lib.call("hv_timestep_embedding", t.data_ptr(), dst.shape[0], dst.shape[1], dst.data_ptr(), stream)


def accumulate_window(lib, stream, pred, rep, Cc, frames, acc, counter):
    f_win, H, W, ldc = pred.shape[0] // rep, pred.shape[1], pred.shape[2], pred.shape[3]
    lib.call("hv_accumulate_window", pred.data_ptr(), ldc, rep, Cc, f_win, H, W, frames.data_ptr(), acc.shape[2],
             acc.data_ptr(), counter.data_ptr(), stream)


def cfg_ddim_step(lib, stream, latents, acc, counter, rep, coeffs):
    """coeffs: device fp32 [5] = {guidance, sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)}."""
    _, Cc, Fr, H, W = latents.shape
    lib.call("hv_cfg_ddim_step", latents.data_ptr(), acc.data_ptr(), counter.data_ptr(), rep, Cc, Fr, H, W,
             coeffs.data_ptr(), stream)
