"""Kernel-level parity cases, shared by the emulator suite (CPU, tiny shapes) and the GPU suite.

Every case builds seeded bf16 inputs, runs ONE C-ABI entry point through `humanvid_amd.ops`, and
compares with a plain PyTorch fp32 CPU computation of the reference op on the same (bf16-rounded)
inputs.  Tolerance: the kernels accumulate in fp32 and round the result to bf16 once, so the bound
is bf16 output rounding (2^-9 relative) plus summation-order noise: NRMSE <= 4e-3 (stated per case).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from humanvid_amd import _abi as A
from humanvid_amd import ops, packing

BF16 = torch.bfloat16
TOL = 4e-3


def nrmse(a: torch.Tensor, ref: torch.Tensor) -> float:
    a, ref = a.detach().float().cpu(), ref.detach().float().cpu()
    return float((a - ref).norm() / (ref.norm() + 1e-12))


def rnd(g, *shape, scale=1.0):
    return torch.randn(*shape, generator=g) * scale


class Ctx:
    def __init__(self, lib, device="cpu", stream=None):
        self.lib, self.device, self.stream = lib, device, stream

    def dev(self, t, dtype=None):
        t = t.to(dtype) if dtype is not None else t
        return t.contiguous().to(self.device)

    def bf(self, t):
        return self.dev(t, BF16)

    def sync(self):
        if self.device != "cpu":
            torch.cuda.synchronize()


def r(t):  # the bf16-rounded value as fp32 (what the kernel actually sees)
    return t.to(BF16).float()


# ----------------------------------------------------------------------------------------- GEMM
def case_gemm(cx: Ctx, M=200, N=320, K=320, seed=0, residual=True, out_f32=False, two_source=False, transposed=False):
    g = torch.Generator().manual_seed(seed)
    x, w = rnd(g, M, K), rnd(g, N, K, scale=K**-0.5)
    bias, res = rnd(g, N, scale=0.1), rnd(g, M, N)
    ref = r(x) @ r(w).t() + bias
    if residual:
        ref = ref + r(res)
    y = torch.zeros(M, N, dtype=torch.float32 if out_f32 else BF16, device=cx.device)
    kw = {}
    xd = cx.bf(x)
    if two_source:
        k1 = (K // 2) // 64 * 64
        x1, x2 = cx.bf(x[:, :k1]), cx.bf(x[:, k1:])
        kw.update(x2=x2, k1=k1)
        xd = x1
    yt = None
    if transposed:
        ns = (N // 2) // 16 * 16
        yt = torch.zeros(N - ns, M, dtype=BF16, device=cx.device)
        kw.update(yt=yt, n_split=ns)
    ops.gemm(cx.lib, cx.stream, xd, cx.bf(w), y, bias=cx.dev(bias), residual=cx.bf(res) if residual else None, **kw)
    cx.sync()
    if transposed:
        ns = kw["n_split"]
        e = max(nrmse(y[:, :ns], ref[:, :ns]), nrmse(yt.t(), ref[:, ns:]))
    else:
        e = nrmse(y, ref)
    assert e < TOL, f"gemm nrmse {e}"
    return e


def case_gemm_row_perm(cx: Ctx, X=6, Y=4, P=24, N=320, K=320, seed=20, form="res"):
    """row-permuted output (+ residual read): row (x*Y + y)*P + p lands at (y*X + x)*P + p.
    form "res": bias + residual read at the DESTINATION row (the frame-sharded motion module's output projection);
    "ln": the LayerNorm fold (row statistics of the LOGICAL input row) + bias, no residual (its QKV projection);
    "res_stats": "res" + the LayerNorm partial statistics of the output (ln_part), which belong to the DESTINATION row."""
    g = torch.Generator().manual_seed(seed)
    M = X * Y * P
    x, w = rnd(g, M, K), rnd(g, N, K, scale=K**-0.5)
    bias, res = rnd(g, N, scale=0.1), rnd(g, M, N)
    idx = torch.arange(M)
    blk, pp = idx // P, idx % P
    dst = ((blk % Y) * X + blk // Y) * P + pp
    ref = torch.zeros(M, N)
    y = torch.zeros(M, N, dtype=BF16, device=cx.device)
    kw = dict(bias=cx.dev(bias), row_perm=(X, Y, P))
    if form == "ln":
        mean, rstd = rnd(g, M, scale=0.3), 0.5 + torch.rand(M, generator=g)
        colsum = r(w).sum(dim=1)
        ref[dst] = rstd[:, None] * (r(x) @ r(w).t() - mean[:, None] * colsum[None, :]) + bias
        kw.update(row_mean=cx.dev(mean), row_rstd=cx.dev(rstd), colsum=cx.dev(colsum))
    else:
        ref[dst] = r(x) @ r(w).t() + bias
        ref = ref + r(res)  # the residual is read at the DESTINATION row
        kw.update(residual=cx.bf(res))
    part = None
    if form == "res_stats":
        np_ = ops.gemm(cx.lib, cx.stream, cx.bf(x), cx.bf(w), y, query_ln_parts=True, **kw)
        if np_ > 0:  # (0: this problem runs on a kernel that leaves no statistics -- nothing to check beyond the output)
            part = torch.full((M, np_, 2), float("nan"), dtype=torch.float32, device=cx.device)
            kw["ln_part"] = part
    ops.gemm(cx.lib, cx.stream, cx.bf(x), cx.bf(w), y, **kw)
    cx.sync()
    e = nrmse(y, ref)
    assert e < TOL, f"gemm row_perm ({form}) nrmse {e}"
    if part is not None:
        got = part.cpu().double().sum(dim=1)  # per DESTINATION row: sum and sum of squares over all columns
        want = torch.stack([ref.double().sum(dim=1), (ref.double() ** 2).sum(dim=1)], dim=1)
        assert torch.isfinite(got).all()
        es = float((got - want).norm() / want.norm())
        assert es < 5e-3, f"gemm row_perm ln_part nrmse {es}"
    return e


def case_gemm_prologue(cx: Ctx, n_img=3, rows=50, N=128, K=320, seed=1):
    """GroupNorm-apply (+SiLU) fused on the A operand: y = silu(x*scale[img]+shift[img]) @ W^T."""
    g = torch.Generator().manual_seed(seed)
    M = n_img * rows
    x, w = rnd(g, M, K), rnd(g, N, K, scale=K**-0.5)
    sc, sh = 1 + 0.3 * rnd(g, n_img, K), 0.2 * rnd(g, n_img, K)
    a = F.silu(r(x).view(n_img, rows, K) * sc[:, None] + sh[:, None]).view(M, K)
    ref = r(a) @ r(w).t()  # the kernel rounds the transformed operand to bf16 before the MFMA
    y = torch.zeros(M, N, dtype=BF16, device=cx.device)
    ops.gemm(cx.lib, cx.stream, cx.bf(x), cx.bf(w), y, pro_scale=cx.dev(sc), pro_shift=cx.dev(sh),
             rows_per_image=rows, pro_act=A.ACT_SILU)
    cx.sync()
    e = nrmse(y, ref)
    assert e < TOL, f"gemm prologue nrmse {e}"
    return e


def case_affine_apply(cx: Ctx, n_img=3, rows=50, C=64, act=A.ACT_NONE, seed=40):
    """GroupNorm apply as its own pass: y = act(x * scale[img] + shift[img]), rounded to bf16 once."""
    g = torch.Generator().manual_seed(seed)
    M = n_img * rows
    x = rnd(g, M, C)
    sc, sh = 1 + 0.3 * rnd(g, n_img, C), 0.2 * rnd(g, n_img, C)
    ref = r(x).view(n_img, rows, C) * sc[:, None] + sh[:, None]
    if act == A.ACT_SILU:
        ref = F.silu(ref)
    y = torch.zeros(M, C, dtype=BF16, device=cx.device)
    ops.affine_apply(cx.lib, cx.stream, cx.bf(x), cx.dev(sc), cx.dev(sh), y, rows_per_image=rows, act=act)
    cx.sync()
    e = nrmse(y, ref.view(M, C))
    assert e < TOL, f"affine_apply nrmse {e}"
    return e


def case_affine_apply_cat(cx: Ctx, n_img=3, rows=50, C1=64, C2=32, seed=43):
    """the same pass over the channel concatenation [x | x2] (decoder ResnetBlock3D input), SiLU, into one [rows, C1 + C2]
    activation -- and what the convolution behind it then computes equals the convolution with the operand prologue"""
    g = torch.Generator().manual_seed(seed)
    M = n_img * rows
    x, x2 = rnd(g, M, C1), rnd(g, M, C2)
    sc, sh = 1 + 0.3 * rnd(g, n_img, C1 + C2), 0.2 * rnd(g, n_img, C1 + C2)
    ref = F.silu(torch.cat([r(x), r(x2)], dim=1).view(n_img, rows, C1 + C2) * sc[:, None] + sh[:, None])
    y = torch.zeros(M, C1 + C2, dtype=BF16, device=cx.device)
    ops.affine_apply(cx.lib, cx.stream, cx.bf(x), cx.dev(sc), cx.dev(sh), y, rows_per_image=rows, act=A.ACT_SILU, x2=cx.bf(x2))
    cx.sync()
    e = nrmse(y, ref.view(M, C1 + C2))
    assert e < TOL, f"affine_apply_cat nrmse {e}"
    return e


def case_fp8_amax_under_command_list_replay(cx: Ctx, n_img=2, Lk=72, H=8, D=40, seed=90):
    """the fp8 pre-pass's per-head V amax is reset by a KERNEL (hv_attention_fp8_zero_kernel through hv_launch), so a
    recorded command list (hv_cmdlist_*: the replay form of a frame-sharded denoising step) re-zeroes it on every replay.
    With the round-2 hipMemsetAsync the reset was not recorded and the atomicMax kept the running maximum of every layer
    and earlier step.  Record phase 1 once, shrink V in place, replay: the amax must follow."""
    import ctypes

    g = torch.Generator().manual_seed(seed)
    Cc = H * D
    k, vt = cx.bf(rnd(g, n_img * Lk, Cc)), cx.bf(rnd(g, Cc, n_img * Lk))
    T = (Lk + 63) // 64
    ks, va = torch.zeros(n_img, H, T, device=cx.device), torch.zeros(H, device=cx.device)
    kw = dict(n_images=n_img, heads=H, D=D, L=Lk, ldk=Cc, ldvt=n_img * Lk, phase=1)
    h = ctypes.c_void_p()
    cx.lib.call("hv_cmdlist_begin")
    try:
        ops.attention_fp8_quantize(cx.lib, cx.stream, k, vt, ks, va, **kw)  # recorded (and executed)
    finally:
        cx.lib.call("hv_cmdlist_end", ctypes.byref(h))
    cx.sync()
    first = va.clone()
    want = vt.float().view(H, D, n_img * Lk).abs().amax(dim=(1, 2))
    assert nrmse(first, want) < 1e-6
    vt.mul_(0.125)  # exact in bf16
    cx.lib.call("hv_cmdlist_run", h, cx.stream)
    cx.sync()
    assert torch.equal(va, first * 0.125), (va, first)
    cx.lib.call("hv_cmdlist_destroy", h)


def case_layernorm_stats(cx: Ctx, M=77, C=320, seed=44, offset=3.0):
    """row statistics of hv_layernorm_stats (mean, 1/sqrt(var + eps)) against torch on the bf16-rounded rows; C = 320 / 640 /
    1280 take the several-rows-per-wave kernel, other widths the one-row-per-wave kernel; M not a multiple of the rows per
    workgroup exercises the dead rows"""
    g = torch.Generator().manual_seed(seed)
    x = rnd(g, M, C) * (1 + torch.arange(C) % 3) + offset
    xr = r(x)
    mean, rstd = torch.zeros(M, device=cx.device), torch.zeros(M, device=cx.device)
    ops.layernorm_stats(cx.lib, cx.stream, cx.bf(x), mean, rstd)
    cx.sync()
    ref_rstd = 1.0 / torch.sqrt(xr.var(dim=1, unbiased=False) + 1e-5)
    e1, e2 = nrmse(mean, xr.mean(dim=1)), nrmse(rstd, ref_rstd)
    assert e1 < 1e-5 and e2 < 1e-5, f"layernorm stats C={C}: mean {e1} rstd {e2}"
    return e1, e2


def case_gemm_lnfold(cx: Ctx, B=2, Fr=3, P=20, C=320, N=192, seed=2):
    """LayerNorm + positional encoding folded into the projection (motion-module QKV)."""
    g = torch.Generator().manual_seed(seed)
    M = B * Fr * P
    x = rnd(g, M, C) + 0.5
    gamma, beta = 1 + 0.2 * rnd(g, C), 0.1 * rnd(g, C)
    w, bias = rnd(g, N, C, scale=C**-0.5), rnd(g, N, scale=0.1)
    pe = rnd(g, Fr, C, scale=0.5)
    xr = r(x)
    ln = F.layer_norm(xr, (C,), gamma, beta, 1e-5)
    frame = (torch.arange(M) // P) % Fr
    ref = (ln + pe[frame]) @ w.t() + bias
    wf, colsum, bfold = packing.fold_layernorm(w, bias, gamma, beta)
    pet = packing.pe_table(pe, w)
    mean = torch.zeros(M, device=cx.device)
    rstd = torch.zeros(M, device=cx.device)
    xd = cx.bf(x)
    ops.layernorm_stats(cx.lib, cx.stream, xd, mean, rstd)
    y = torch.zeros(M, N, dtype=BF16, device=cx.device)
    ops.gemm(cx.lib, cx.stream, xd, cx.dev(wf), y, bias=cx.dev(bfold), row_mean=mean, row_rstd=rstd,
             colsum=cx.dev(colsum), pe=cx.dev(pet), pe_period=P, pe_frames=Fr)
    cx.sync()
    e_stats = nrmse(mean, xr.mean(1))
    e = nrmse(y, ref)
    assert e_stats < 1e-5, f"ln mean {e_stats}"
    assert e < 6e-3, f"ln-fold nrmse {e}"  # folded weights are rounded once more than the reference's
    return e


def case_gemm_forms(cx: Ctx, M=520, C=128, N=192, P=128, form="ln", seed=30, return_output=False, res_rowvec=True):
    """The epilogue forms of the denoising path as the engine launches them (hv_gemm_epilogue_fast on the LDS-DMA kernel
    when M >= 256 and the per-row table period P is a multiple of the wave sub-tile):
      ln        LayerNorm fold + bias + positional-encoding row (motion-module QKV)
      ln_yt     LayerNorm fold + bias, last third stored transposed (spatial QKV, V^T tail)
      ln_geglu  LayerNorm fold + bias + GEGLU, no residual (feed-forward input projection)
      res       bias + per-batch row vector + residual, in place (attention / feed-forward output projections)
      plain     bias only"""
    g = torch.Generator().manual_seed(seed)
    x = rnd(g, M, C) + 0.25
    xr = r(x)
    xd = cx.bf(x)
    gamma, beta = 1 + 0.2 * rnd(g, C), 0.1 * rnd(g, C)
    if form in ("ln", "ln_yt", "ln_geglu"):
        n_w = 2 * N if form == "ln_geglu" else N
        w, bias = rnd(g, n_w, C, scale=C**-0.5), rnd(g, n_w, scale=0.1)
        ln = F.layer_norm(xr, (C,), gamma, beta, 1e-5)
        wf, colsum, bfold = packing.fold_layernorm(w, bias, gamma, beta)
        mean, rstd = torch.zeros(M, device=cx.device), torch.zeros(M, device=cx.device)
        ops.layernorm_stats(cx.lib, cx.stream, xd, mean, rstd)
        kw = dict(row_mean=mean, row_rstd=rstd)
        if form == "ln":
            Fr = 3
            pe = rnd(g, Fr, C, scale=0.5)
            frame = (torch.arange(M) // P) % Fr
            ref = (ln + pe[frame]) @ w.t() + bias
            y = torch.zeros(M, N, dtype=BF16, device=cx.device)
            ops.gemm(cx.lib, cx.stream, xd, cx.dev(wf), y, bias=cx.dev(bfold), colsum=cx.dev(colsum),
                     pe=cx.dev(packing.pe_table(pe, w)), pe_period=P, pe_frames=Fr, **kw)
            cx.sync()
            e = nrmse(y, ref)
        elif form == "ln_yt":
            ref = ln @ w.t() + bias
            ns = (2 * N // 3) // 64 * 64
            y = torch.zeros(M, ns, dtype=BF16, device=cx.device)
            yt = torch.zeros(N - ns, M, dtype=BF16, device=cx.device)
            ops.gemm(cx.lib, cx.stream, xd, cx.dev(wf), y, bias=cx.dev(bfold), colsum=cx.dev(colsum), yt=yt, n_split=ns, **kw)
            cx.sync()
            e = max(nrmse(y, ref[:, :ns]), nrmse(yt.t(), ref[:, ns:]))
        else:
            h, gate = (ln @ w.t() + bias).chunk(2, dim=-1)
            ref = h * F.gelu(gate)
            order = packing.geglu_row_order(n_w, wf.device)
            y = torch.zeros(M, N, dtype=BF16, device=cx.device)
            ops.gemm(cx.lib, cx.stream, xd, cx.dev(wf[order].contiguous()), y, bias=cx.dev(bfold[order].contiguous()),
                     colsum=cx.dev(colsum[order].contiguous()), geglu=True, **kw)
            cx.sync()
            e = nrmse(y, ref)
        tol = 6e-3  # folded weights are rounded once more than the reference's
    else:
        w, bias = rnd(g, N, C, scale=C**-0.5), rnd(g, N, scale=0.1)
        ref = xr @ r(w).t() + bias
        y = torch.zeros(M, N, dtype=BF16, device=cx.device)
        if form == "res":
            nb = (M + P - 1) // P
            rv = rnd(g, nb, N, scale=0.3)
            res = rnd(g, M, N)
            ref = ref + r(res)
            y.copy_(cx.bf(res))  # the residual stream is updated in place
            if res_rowvec:
                ref = ref + rv[torch.arange(M) // P]
                ops.gemm(cx.lib, cx.stream, xd, cx.bf(w), y, bias=cx.dev(bias), rowvec=cx.dev(rv), rowvec_period=P, residual=y)
            else:  # the feed-forward output projection: bias + residual only
                ops.gemm(cx.lib, cx.stream, xd, cx.bf(w), y, bias=cx.dev(bias), residual=y)
        else:
            ops.gemm(cx.lib, cx.stream, xd, cx.bf(w), y, bias=cx.dev(bias))
        cx.sync()
        e = nrmse(y, ref)
        tol = TOL
    assert e < tol, f"gemm form {form} nrmse {e}"
    if return_output:  # (for bit-wise comparisons of kernel selections; the transposed tail of ln_yt rides along)
        return torch.cat([y.float().cpu().reshape(-1), yt.float().cpu().reshape(-1)]) if form == "ln_yt" else y.float().cpu()
    return e


def case_gemm_geglu(cx: Ctx, M=70, C=64, seed=3):
    g = torch.Generator().manual_seed(seed)
    x = rnd(g, M, C)
    w, b = rnd(g, 8 * C, C, scale=C**-0.5), rnd(g, 8 * C, scale=0.1)
    res = rnd(g, M, 4 * C)
    h, gate = (r(x) @ r(w).t() + b).chunk(2, dim=-1)
    ref = h * F.gelu(gate) + r(res)
    wp, bp, _ = packing.pack_geglu(w, b)
    y = torch.zeros(M, 4 * C, dtype=BF16, device=cx.device)
    ops.gemm(cx.lib, cx.stream, cx.bf(x), cx.bf(wp), y, bias=cx.dev(bp), geglu=True, residual=cx.bf(res))
    cx.sync()
    e = nrmse(y, ref)
    assert e < TOL, f"geglu nrmse {e}"
    return e


def case_geglu_pointwise(cx: Ctx, M=4096, C=64, lo=-9.0, hi=9.0):
    """the epilogue's GELU (hv_gelu_times: erfc as 2^(u q(u))) pointwise against F.gelu (exact erf form, the reference's
    GEGLU.gelu) over [lo, hi]: value rows are 1, gate rows pass x[:, 0] through, so y[m, :] = bf16(gelu(x[m, 0])).
    Tolerance: one bf16 rounding of the result (2^-8 relative) + 1e-6 absolute."""
    gate = torch.linspace(lo, hi, M).to(BF16).float()  # bf16-exact gate values
    x = torch.zeros(M, C)
    x[:, 0], x[:, 1] = gate, 1.0
    w = torch.zeros(8 * C, C)
    w[: 4 * C, 1] = 1.0   # value half: h = 1
    w[4 * C :, 0] = 1.0   # gate half: g = x[:, 0]
    wp, bp, _ = packing.pack_geglu(w, torch.zeros(8 * C))
    y = torch.zeros(M, 4 * C, dtype=BF16, device=cx.device)
    ops.gemm(cx.lib, cx.stream, cx.bf(x), cx.bf(wp), y, bias=cx.dev(bp), geglu=True)
    cx.sync()
    ref = F.gelu(gate.double())[:, None].expand(M, 4 * C)
    err = (y.double().cpu() - ref).abs()
    bound = ref.abs() * 2.0**-8 + 1e-6
    worst = float((err / bound).max())
    assert worst <= 1.0, f"geglu pointwise: worst error / bound = {worst}"
    return worst


# ----------------------------------------------------------------------------------------- conv
def case_conv(cx: Ctx, n=2, H=12, W=20, C1=32, C2=0, Cout=40, mode=A.CONV_S1, pro=True, temb=True, residual=True,
              out_act=A.ACT_NONE, seed=4, check=None, return_output=False):
    """check: image indices the CPU reference is evaluated on (None = all); the kernel always runs all n images."""
    g = torch.Generator().manual_seed(seed)
    Cin = C1 + C2
    x = rnd(g, n, Cin, H, W)
    w, bias = rnd(g, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5), rnd(g, Cout, scale=0.1)
    sc, sh = 1 + 0.3 * rnd(g, n, Cin), 0.2 * rnd(g, n, Cin)
    idx = list(range(n)) if check is None else list(check)
    a = r(x[idx])
    if pro:
        a = r(F.silu(a * sc[idx][:, :, None, None] + sh[idx][:, :, None, None]))
    if mode == A.CONV_UP2:
        a = F.interpolate(a, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(a, r(w), bias, stride=2 if mode == A.CONV_S2 else 1, padding=1)
    Ho, Wo = ref.shape[2], ref.shape[3]
    tv = rnd(g, 1, Cout, scale=0.3)
    if temb:
        ref = ref + tv[:, :, None, None]
    res = rnd(g, 1, Cout, Ho, Wo)
    if residual:
        ref = ref + r(res)
    if out_act == A.ACT_SILU:
        ref = F.silu(ref)
    xh = x.permute(0, 2, 3, 1)
    x1 = cx.bf(xh[..., :C1])
    x2 = cx.bf(xh[..., C1:]) if C2 else None
    y = torch.zeros(n, Ho, Wo, Cout, dtype=BF16, device=cx.device)
    ops.conv3x3(cx.lib, cx.stream, x1, cx.dev(packing.pack_conv3x3(w, Cin)), y, x2=x2, mode=mode,
                pro_scale=cx.dev(sc) if pro else None, pro_shift=cx.dev(sh) if pro else None,
                pro_act=A.ACT_SILU if pro else A.ACT_NONE, bias=cx.dev(bias),
                rowvec=cx.dev(tv) if temb else None, images_per_rowvec=n,
                residual=cx.bf(res.permute(0, 2, 3, 1)) if residual else None, out_act=out_act)
    cx.sync()
    e = nrmse(y[idx].permute(0, 3, 1, 2), ref)
    assert e < TOL, f"conv mode {mode} nrmse {e}"
    if return_output:  # (bit-wise comparisons of kernel selections)
        return y.float().cpu()
    return e


# ----------------------------------------------------------------------------------------- norms
def case_groupnorm(cx: Ctx, n=3, H=6, W=10, C1=320, C2=0, groups=32, seed=5, offset=0.7, spread=1.0, splits=3):
    """offset >> spread is the cancellation case of a sum / sum-of-squares variance (|mean| >> std)."""
    g = torch.Generator().manual_seed(seed)
    C = C1 + C2
    x = rnd(g, n, C, H, W) * (1 + torch.arange(C).view(1, C, 1, 1) % 5) * spread + offset
    gamma, beta = 1 + 0.2 * rnd(g, C), 0.1 * rnd(g, C)
    xr = r(x)
    ref = F.group_norm(xr, groups, gamma, beta, 1e-5)
    xh = x.permute(0, 2, 3, 1)
    partial = torch.zeros(n * 64 * groups * 2, device=cx.device)
    scale, shift = torch.zeros(n, C, device=cx.device), torch.zeros(n, C, device=cx.device)
    ops.groupnorm_affine(cx.lib, cx.stream, cx.bf(xh[..., :C1]), cx.dev(gamma), cx.dev(beta), groups, 1e-5, partial,
                         scale, shift, x2=cx.bf(xh[..., C1:]) if C2 else None, splits=splits)
    cx.sync()
    got = xr * scale.cpu()[:, :, None, None] + shift.cpu()[:, :, None, None]
    e = nrmse(got, ref)
    assert e < 1e-4, f"groupnorm nrmse {e}"
    return e


def case_gn_parts_conv(cx: Ctx, n=3, H=16, W=16, Cin=64, Cout=320, C2=0, groups=32, seed=51, offset=0.4, mode=None):
    """GroupNorm statistics emitted by the PRODUCING convolution (hv_conv3x3 gn_part) -> hv_groupnorm_from_parts must give
    the scale / shift that a statistics pass over the stored output gives (hv_groupnorm_affine) and that torch computes from
    the stored bf16 values; `offset` shifts the output mean away from zero (the cancellation case of sum / sum-of-squares).
    C2 > 0: a second, independently produced activation is concatenated (the up-block norm1: groups straddle the seam)."""
    g = torch.Generator().manual_seed(seed)
    mode = A.CONV_S1 if mode is None else mode
    Ho, Wo = (H, W) if mode == A.CONV_S1 else ((H + 1) // 2, (W + 1) // 2) if mode == A.CONV_S2 else (2 * H, 2 * W)
    outs, parts = [], []
    for co in ([Cout] if C2 == 0 else [Cout, C2]):
        x = rnd(g, n, H, W, Cin)
        wt = rnd(g, co, Cin, 3, 3, scale=(9 * Cin) ** -0.5)
        bias = rnd(g, co, scale=0.1) + offset
        y = torch.zeros(n, Ho, Wo, co, dtype=BF16, device=cx.device)
        kw = dict(mode=mode, bias=cx.dev(bias))
        xd, wd = cx.bf(x), cx.bf(packing.pack_conv3x3(wt))
        np_ = ops.conv3x3(cx.lib, cx.stream, xd, wd, y, query_gn_parts=True, **kw)
        assert np_ > 0
        part = torch.zeros(n, np_, co, 2, device=cx.device)
        ops.conv3x3(cx.lib, cx.stream, xd, wd, y, gn_part=part, **kw)
        outs.append(y)
        parts.append(part)
    C = Cout + C2
    gamma, beta = 1 + 0.2 * rnd(g, C), 0.1 * rnd(g, C)
    sc1, sh1 = torch.zeros(n, C, device=cx.device), torch.zeros(n, C, device=cx.device)
    sc2, sh2 = torch.zeros(n, C, device=cx.device), torch.zeros(n, C, device=cx.device)
    ops.groupnorm_from_parts(cx.lib, cx.stream, parts[0], cx.dev(gamma), cx.dev(beta), groups, 1e-5, Ho * Wo, sc1, sh1,
                             part2=parts[1] if C2 else None)
    partial = torch.zeros(n * 64 * groups * 2, device=cx.device)
    ops.groupnorm_affine(cx.lib, cx.stream, outs[0], cx.dev(gamma), cx.dev(beta), groups, 1e-5, partial, sc2, sh2,
                         x2=outs[1] if C2 else None, splits=min(4, max(1, Ho * Wo // 48)))
    cx.sync()
    ycat = torch.cat([o.float().cpu() for o in outs], dim=3).permute(0, 3, 1, 2)
    ref = F.group_norm(ycat, groups, gamma, beta, 1e-5)
    got1 = ycat * sc1.cpu()[:, :, None, None] + sh1.cpu()[:, :, None, None]
    got2 = ycat * sc2.cpu()[:, :, None, None] + sh2.cpu()[:, :, None, None]
    e1, e2 = nrmse(got1, ref), nrmse(got2, ref)
    # the fused statistics see the fp32 values before the bf16 store: the difference is the store's rounding noise
    assert e1 < 2e-3 and e2 < 1e-4, f"groupnorm from conv parts nrmse {e1} (statistics pass: {e2})"
    return e1


def case_gn_parts_gemm(cx: Ctx, n=4, rows=128, C=320, K=128, groups=32, seed=52, offset=0.3, part_rows=64):
    """the same for the projection-out GEMM (bias + residual in place, 128x128x64 LDS-DMA kernel, permuted epilogue)"""
    g = torch.Generator().manual_seed(seed)
    M = n * rows
    x, wt = rnd(g, M, K), rnd(g, C, K, scale=K**-0.5)
    bias, res = rnd(g, C, scale=0.1) + offset, rnd(g, M, C)
    y = cx.bf(res)
    xd, wd = cx.bf(x), cx.bf(wt)
    kw = dict(bias=cx.dev(bias), residual=y)
    np_ = ops.gemm(cx.lib, cx.stream, xd, wd, y, gn_rows_per_image=rows, query_gn_parts=True, **kw)
    assert np_ == rows // part_rows, np_  # one partial sum per wave sub-tile: 64 rows (128 x 128 kernel), 32 (wide kernel)
    # the part table is followed by a sentinel tail: a wave sub-tile beyond the ragged M edge (M % 128 == 64 on the 128 x 128
    # kernel: n * rows / 64 odd) must not write a part of an image that does not exist
    store = torch.full((n * np_ * C * 2 + 2 * C * 2,), -777.0, device=cx.device)
    part = store[:n * np_ * C * 2].view(n, np_, C, 2)
    part.zero_()
    ops.gemm(cx.lib, cx.stream, xd, wd, y, gn_part=part, gn_rows_per_image=rows, **kw)
    cx.sync()
    assert bool((store[n * np_ * C * 2:] == -777.0).all()), "gn_part written beyond the last image"
    gamma, beta = 1 + 0.2 * rnd(g, C), 0.1 * rnd(g, C)
    sc, sh = torch.zeros(n, C, device=cx.device), torch.zeros(n, C, device=cx.device)
    ops.groupnorm_from_parts(cx.lib, cx.stream, part, cx.dev(gamma), cx.dev(beta), groups, 1e-5, rows, sc, sh)
    cx.sync()
    want = r(x) @ r(wt).t() + bias + r(res)
    assert nrmse(y, want) < TOL
    y4 = y.float().cpu().view(n, rows, C).permute(0, 2, 1)[..., None]  # [n, C, rows, 1]
    ref = F.group_norm(y4, groups, gamma, beta, 1e-5)
    got = y4 * sc.cpu()[:, :, None, None] + sh.cpu()[:, :, None, None]
    e = nrmse(got, ref)
    assert e < 2e-3, f"groupnorm from gemm parts nrmse {e}"
    # a problem whose kernel cannot emit statistics reports 0 parts and refuses gn_part (rows per image not whole wave blocks: 96 for 64-row blocks, 48 for 32-row blocks)
    assert ops.gemm(cx.lib, cx.stream, xd, wd, y, gn_rows_per_image=3 * part_rows // 2, query_gn_parts=True, **kw) == 0
    return e


def case_ln_parts_gemm(cx: Ctx, M=700, C=320, K=128, seed=59, offset=0.5, residual=True):
    """LayerNorm row statistics emitted by the PRODUCING GEMM (hv_gemm ln_part) -> hv_layernorm_from_parts must give the
    mean / rstd that the statistics pass over the stored output gives (hv_layernorm_stats); ragged M."""
    g = torch.Generator().manual_seed(seed)
    x, wt = rnd(g, M, K), rnd(g, C, K, scale=K**-0.5)
    bias, res = rnd(g, C, scale=0.1) + offset, rnd(g, M, C)
    y = cx.bf(res) if residual else torch.zeros(M, C, dtype=BF16, device=cx.device)
    xd, wd = cx.bf(x), cx.bf(wt)
    kw = dict(bias=cx.dev(bias))
    if residual:
        kw["residual"] = y
    np_ = ops.gemm(cx.lib, cx.stream, xd, wd, y, query_ln_parts=True, **kw)
    assert np_ in (C // 64, 4), np_  # (64-column blocks from the tile kernels, one part per wave from hv_gemm_wr_kernel)
    part = torch.zeros(M, np_, 2, device=cx.device)
    ops.gemm(cx.lib, cx.stream, xd, wd, y, ln_part=part, **kw)
    m1, r1 = torch.zeros(M, device=cx.device), torch.zeros(M, device=cx.device)
    m2, r2 = torch.zeros(M, device=cx.device), torch.zeros(M, device=cx.device)
    ops.layernorm_from_parts(cx.lib, cx.stream, part, C, m1, r1)
    ops.layernorm_stats(cx.lib, cx.stream, y, m2, r2)
    cx.sync()
    yf = y.float().cpu()
    mean, rstd = yf.mean(1), (yf.var(1, unbiased=False) + 1e-5).rsqrt()
    for got_m, got_r, tol in ((m1, r1, 2e-3), (m2, r2, 1e-4)):
        em = float((got_m.cpu() - mean).abs().max() / yf.std())
        er = float(((got_r.cpu() - rstd) / rstd).abs().max())
        assert em < tol and er < tol, (em, er)
    return 0.0


# ----------------------------------------------------------------------------------------- attention
def case_attention(cx: Ctx, D=40, n_img=4, Lq=72, Lb=40, seed=6, check=None, q_stride=1, spike=False,
                   fp8=False):
    """images [0, n/2) are CFG-unconditional (own keys only); the rest append bank batch 1.
    check / q_stride: images and query rows the CPU reference is evaluated on (the kernel runs everything).
    spike: keys in LATER tiles (own key 70+, bank key 3) are made strongly aligned with a few queries, so that those
    queries' running maximum jumps far past the deferred-rescale threshold in the middle of the key loop (the branch
    random data never takes: cdna_hip_programming.md rule 26)."""
    g = torch.Generator().manual_seed(seed)
    H = 8
    Cc = H * D
    q, k, v = rnd(g, n_img, Lq, Cc), rnd(g, n_img, Lq, Cc), rnd(g, n_img, Lq, Cc)
    kb, vb = rnd(g, 2, Lb, Cc), rnd(g, 2, Lb, Cc)
    if spike:
        # spike = True: scores 30-45 (exp2 domain) above the rest -- past the deferred-rescale threshold 2^8, inside fp32 range;
        # spike = 8.0 (a factor): 200-360 above -- exp2 overflows against the first tile's reference, which is what sends
        # the optimistic head-dim-40 kernel (hv_attention40.h) into its careful second pass
        # (with a factor only ONE own key per image is spiked: several keys with logits in the hundreds would compete through
        #  cross-terms whose bf16 rounding -- 2^-9 of a logit of ~100 -- is no longer small against their gaps; that is a
        #  property of bf16 logits, not of the branch under test)
        amp = 1.0 if spike is True else float(spike)
        for i in range(n_img):
            for j, ks in enumerate(range(min(Lq - 1, 70), Lq, 97)):
                if amp != 1.0 and j > 0:
                    break
                k[i, ks] = amp * (3.0 + j % 3) * q[i, (5 + 11 * j) % Lq]
        kb[1, min(3, Lb - 1)] = amp * 4.0 * q[n_img - 1, 9 % Lq]
    sel = torch.tensor([-1] * (n_img // 2) + [1] * (n_img - n_img // 2), dtype=torch.int32)

    def heads(t):
        return r(t).view(t.shape[0], t.shape[1], H, D).transpose(1, 2)

    idx = list(range(n_img)) if check is None else list(check)
    ref = torch.zeros(len(idx), len(range(0, Lq, q_stride)), Cc)
    for j, i in enumerate(idx):
        kk, vv = heads(k[i : i + 1]), heads(v[i : i + 1])
        if sel[i] >= 0:
            s = int(sel[i])
            kk = torch.cat([kk, heads(kb[s : s + 1])], dim=2)
            vv = torch.cat([vv, heads(vb[s : s + 1])], dim=2)
        o = F.scaled_dot_product_attention(heads(q[i : i + 1, ::q_stride]), kk, vv)
        ref[j] = o.transpose(1, 2).reshape(-1, Cc)
    o = torch.zeros(n_img * Lq, Cc, dtype=BF16, device=cx.device)
    if True:
        qkv = cx.bf(torch.cat([q, k], dim=-1).view(n_img * Lq, 2 * Cc))  # q | k interleaved rows, ld = 2C
        vt = cx.bf(v.reshape(n_img * Lq, Cc).t())  # [C][n*Lq]
        k2 = cx.bf(kb.reshape(2 * Lb, Cc))
        vt2 = cx.bf(vb.reshape(2 * Lb, Cc).t())
        kw = dict(n_images=n_img, heads=H, D=D, Lq=Lq, L1=Lq, ldq=2 * Cc, ldk=2 * Cc, ldvt=n_img * Lq, ldo=Cc, k2=k2, vt2=vt2,
                  ldk2=Cc, ldvt2=2 * Lb, L2=Lb, bank_sel=cx.dev(sel))
        if fp8:  # e4m3 QK^T / PV: amax both key sources, quantise both with the common V scale, then the fp8 kernel
            T1, T2 = (Lq + 63) // 64, (Lb + 63) // 64
            dev = cx.device
            ks1, va1 = torch.zeros(n_img, H, T1, device=dev), torch.zeros(H, device=dev)
            ks2, va2 = torch.zeros(2, H, T2, device=dev), torch.zeros(H, device=dev)
            k8 = torch.zeros(n_img * Lq, Cc, dtype=torch.uint8, device=dev)
            vt8 = torch.zeros(Cc, n_img * Lq, dtype=torch.uint8, device=dev)
            k28 = torch.zeros(2 * Lb, Cc, dtype=torch.uint8, device=dev)
            vt28 = torch.zeros(Cc, 2 * Lb, dtype=torch.uint8, device=dev)
            own = dict(n_images=n_img, heads=H, D=D, L=Lq, ldk=2 * Cc, ldvt=n_img * Lq)
            bnk = dict(n_images=2, heads=H, D=D, L=Lb, ldk=Cc, ldvt=2 * Lb)
            ops.attention_fp8_quantize(cx.lib, cx.stream, qkv[:, Cc:], vt, ks1, va1, phase=1, **own)
            ops.attention_fp8_quantize(cx.lib, cx.stream, k2, vt2, ks2, va2, phase=1, **bnk)
            ops.attention_fp8_quantize(cx.lib, cx.stream, qkv[:, Cc:], vt, ks1, va1, phase=2, vfloor=va2, k8=k8, vt8=vt8, **own)
            ops.attention_fp8_quantize(cx.lib, cx.stream, k2, vt2, ks2, va2, phase=2, vfloor=va1, k8=k28, vt8=vt28, **bnk)
            kw8 = dict(kw, ldk=Cc, k2=k28, vt2=vt28)
            ops.attention_fp8(cx.lib, cx.stream, qkv, k8, vt8, o, ks1, va1, kscale2=ks2, vamax2=va2, **kw8)
            cx.sync()
            amax = r(k).view(n_img, Lq, H, D)[:, :min(Lq, 64)].abs().amax(dim=(1, 3))  # first tile of every (image, head)
            assert nrmse(ks1[:, :, 0], amax / 384.0) < 1e-6, "fp8 K scale"
            assert nrmse(va1, r(v).view(n_img, Lq, H, D).abs().amax(dim=(0, 1, 3))) < 1e-6, "fp8 V amax"
        else:
            ops.attention(cx.lib, cx.stream, qkv, qkv[:, Cc:], vt, o, **kw)
    cx.sync()
    assert torch.isfinite(o.float()).all()
    e = nrmse(o.view(n_img, Lq, Cc)[idx][:, ::q_stride], ref)
    # bf16: P is rounded to bf16 before P.V (as SDPA kernels do).
    # fp8: all four operands carry e4m3's 3 mantissa bits (rms rounding error 2^-4 / sqrt(3) = 3.6e-2 each).  On i.i.d. random
    # q / k / v the output is an average of independent values, so relative probability errors show up undiminished in it:
    # expected ~ sqrt(2 * 0.036^2 [QK^T] + 0.036^2 [P] + 0.036^2 [V]) ~ 7e-2 worst case, measured 4-5e-2; the bound on the
    # DENOISER's output with fp8 attention (residual stream, structured activations) is the one that matters and is stated
    # and tested separately (tests/test_gpu_fullwidth.py::test_fp8_attention_forward: <= 3e-2 against the fp32 oracle).
    # (measured 4.9e-2 ... 5.5e-2 on MI355X at every tested geometry; the bound is that + 20 %)
    tol = 6.6e-2 if fp8 else 6e-3
    assert e < tol, f"attention D={D} fp8={fp8} nrmse {e}"
    return e


def case_temporal(cx: Ctx, D=40, B=2, Fr=5, P=6, seed=7):
    g = torch.Generator().manual_seed(seed)
    H = 8
    Cc = H * D
    qkv = rnd(g, B * Fr * P, 3 * Cc)
    t = r(qkv).view(B, Fr, P, 3, H, D)
    q, k, v = (t[:, :, :, i].permute(0, 2, 3, 1, 4).reshape(B * P, H, Fr, D) for i in range(3))
    o = F.scaled_dot_product_attention(q, k, v)  # [(b p), h, f, d]
    ref = o.view(B, P, H, Fr, D).permute(0, 3, 1, 2, 4).reshape(B * Fr * P, Cc)
    out = torch.zeros(B * Fr * P, Cc, dtype=BF16, device=cx.device)
    ops.temporal_attention(cx.lib, cx.stream, cx.bf(qkv), out, B=B, F=Fr, P=P, heads=H, D=D)
    cx.sync()
    e = nrmse(out, ref)
    assert e < TOL, f"temporal D={D} nrmse {e}"
    # frame-sharded form: 'ranks' shards of Fq frames each query against the gathered K/V
    if Fr % 2 == 0:
        ranks, Fq = 2, Fr // 2
        t5 = qkv.view(B, Fr, P, 3 * Cc)
        kvg = torch.stack([t5[:, rk * Fq:(rk + 1) * Fq, :, Cc:] for rk in range(ranks)]).contiguous()  # [R,B,Fq,P,2C]
        for rk in range(ranks):
            ql = cx.bf(t5[:, rk * Fq:(rk + 1) * Fq, :, :Cc].reshape(B * Fq * P, Cc))
            ol = torch.zeros(B * Fq * P, Cc, dtype=BF16, device=cx.device)
            ops.temporal_attention_sharded(cx.lib, cx.stream, ql, cx.bf(kvg), ol, B=B, Fq=Fq, ranks=ranks, P=P,
                                           heads=H, D=D)
            cx.sync()
            want = ref.view(B, Fr, P, Cc)[:, rk * Fq:(rk + 1) * Fq].reshape(B * Fq * P, Cc)
            e2 = nrmse(ol, want)
            assert e2 < TOL, f"temporal sharded rank {rk} nrmse {e2}"
    # all-to-all form: every operand in the [rank][b][F/ranks][P] chunk layout, output in the same order
    if Fr % 2 == 0:
        ranks, Fl = 2, Fr // 2
        t5 = qkv.view(B, ranks, Fl, P, 3 * Cc)
        recv = cx.bf(t5.permute(1, 0, 2, 3, 4).contiguous())  # [R,B,Fl,P,3C]
        osend = torch.zeros(ranks, B, Fl, P, Cc, dtype=BF16, device=cx.device)
        ops.temporal_attention_exchanged(cx.lib, cx.stream, recv, osend, B=B, F_local=Fl, ranks=ranks, P=P, heads=H, D=D)
        cx.sync()
        want = ref.view(B, ranks, Fl, P, Cc).permute(1, 0, 2, 3, 4)
        e3 = nrmse(osend, want)
        assert e3 < TOL, f"temporal exchanged form nrmse {e3}"
    return e


# ----------------------------------------------------------------------------------------- elementwise
def case_elementwise(cx: Ctx, seed=8):
    g = torch.Generator().manual_seed(seed)
    B, Cc, Fr, H, W = 1, 4, 3, 6, 5
    lat = rnd(g, B, Cc, Fr, H, W)
    dst = torch.zeros(2 * B * Fr, H, W, 32, dtype=BF16, device=cx.device)
    ops.pack_ncfhw(cx.lib, cx.stream, cx.dev(lat), dst, rep=2)
    cx.sync()
    want = r(lat).permute(0, 2, 3, 4, 1).reshape(B * Fr, H, W, Cc)
    got = dst.float().cpu()
    assert torch.equal(got[: B * Fr, ..., :Cc], want) and torch.equal(got[B * Fr :, ..., :Cc], want)
    assert float(got[..., Cc:].abs().max()) == 0.0
    back = torch.zeros(2 * B, Cc, Fr, H, W, device=cx.device)
    ops.unpack_nhwc(cx.lib, cx.stream, dst, back)
    cx.sync()
    assert torch.equal(back.cpu()[0], r(lat)[0]) and torch.equal(back.cpu()[1], r(lat)[0])

    src = rnd(g, 1, 6, 2, 16, 8)
    un = torch.zeros(2, 2, 1, 6 * 64, dtype=BF16, device=cx.device)
    ops.pixel_unshuffle(cx.lib, cx.stream, cx.dev(src), un, 8)
    cx.sync()
    ref = F.pixel_unshuffle(src.permute(0, 2, 1, 3, 4).reshape(2, 6, 16, 8), 8).permute(0, 2, 3, 1)
    assert torch.equal(un.float().cpu(), r(ref))

    # Pluecker map generated inside the unshuffle (ray_condition + PixelUnshuffle fused), vs the oracle's
    # ray_condition restatement (pinned against the reference in oracle/gen_golden.py) + torch pixel_unshuffle
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    import oracle_torch as O  # test infrastructure: the checker

    Fc, Hc, Wc = 3, 16, 24
    Kc = torch.tensor([[20.0, 22.0, 12.0, 8.0], [25.0, 21.0, 11.5, 8.5], [18.0, 30.0, 12.5, 7.5]])
    c2w = torch.eye(4).repeat(Fc, 1, 1)
    c2w[1, :3, :3] = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    c2w[1, :3, 3] = torch.tensor([0.3, -0.2, 0.5])
    c2w[2, :3, 3] = torch.tensor([-1.0, 0.25, 0.1])
    un = torch.zeros(Fc, Hc // 8, Wc // 8, 6 * 64, dtype=BF16, device=cx.device)
    ops.plucker_unshuffle(cx.lib, cx.stream, cx.dev(Kc), cx.dev(c2w), Hc, Wc, 8, un)
    cx.sync()
    pl = O.ray_condition(Kc[None], c2w[None], Hc, Wc)[0].permute(0, 3, 1, 2)  # [F,6,H,W]
    ref = F.pixel_unshuffle(pl, 8).permute(0, 2, 3, 1)
    assert float((un.float().cpu() - ref).abs().max()) < 2e-2 and float((un.float().cpu() - ref).norm() / ref.norm()) < 4e-3

    t = torch.tensor([601.0, 32.0])
    te = torch.zeros(2, 320, dtype=BF16, device=cx.device)
    ops.timestep_embedding(cx.lib, cx.stream, cx.dev(t), te)
    cx.sync()
    half = 160
    freq = torch.exp(-math.log(10000.0) * torch.arange(half) / half)
    ang = t[:, None] * freq[None]
    ref = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)
    assert float((te.float().cpu() - ref).abs().max()) < 1e-2  # bf16 output, fp32 range reduction

    # window accumulation + CFG + DDIM
    pred = rnd(g, 2 * 2, H, W, 4)  # rep=2, f_win=2
    frames = torch.tensor([2, 0], dtype=torch.int32)
    acc = torch.zeros(2, Cc, Fr, H, W, device=cx.device)
    cnt = torch.zeros(Fr, device=cx.device)
    ops.accumulate_window(cx.lib, cx.stream, cx.bf(pred), 2, Cc, cx.dev(frames), acc, cnt)
    ops.accumulate_window(cx.lib, cx.stream, cx.bf(pred), 2, Cc, cx.dev(torch.tensor([1, 2], dtype=torch.int32)), acc, cnt)
    cx.sync()
    pr = r(pred).view(2, 2, H, W, 4).permute(0, 4, 1, 2, 3)  # [rep][c][fw][h][w]
    racc = torch.zeros(2, Cc, Fr, H, W)
    racc[:, :, [2, 0]] += pr
    racc[:, :, [1, 2]] += pr
    assert torch.allclose(acc.cpu(), racc) and cnt.cpu().tolist() == [1.0, 1.0, 2.0]
    latd = cx.dev(lat.clone())
    sa, s1a, sap, s1ap, gs = 0.6, 0.8, 0.8, 0.6, 3.5
    ops.cfg_ddim_step(cx.lib, cx.stream, latd, acc, cnt, 2, cx.dev(torch.tensor([gs, sa, s1a, sap, s1ap])))
    cx.sync()
    nz = racc / torch.tensor([1.0, 1.0, 2.0]).view(1, 1, Fr, 1, 1)
    v = nz[0] + gs * (nz[1] - nz[0])
    x0 = sa * lat[0] - s1a * v
    eps = sa * v + s1a * lat[0]
    ref = sap * x0 + s1ap * eps
    assert torch.allclose(latd.cpu()[0], ref, atol=1e-5)
    assert float(acc.abs().max()) == 0.0 and float(cnt.abs().max()) == 0.0
    return 0.0
