"""Kernel parity on a real MI355X through the C ABI (libhumanvid_hip.so), larger shapes than the
emulator suite, including the config-#3 geometry of each kernel family."""
import pytest
import torch

import kernel_cases as kc
from humanvid_amd import _abi as A

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cx():
    from humanvid_amd import lib

    lib.require_gpu()
    return kc.Ctx(lib.load(), "cuda", lib.current_stream())


def test_gemm(cx):
    kc.case_gemm(cx, M=3000, N=960, K=320)
    kc.case_gemm(cx, M=777, N=1284, K=1280, residual=False, out_f32=True)
    kc.case_gemm(cx, M=1024, N=640, K=1920, two_source=True)
    kc.case_gemm(cx, M=1536, N=1920, K=640, transposed=True)


def test_gemm_row_permutation(cx):
    kc.case_gemm_row_perm(cx, X=48, Y=8, P=96, N=960, K=320)   # frame-sharded QKV -> all-to-all send layout
    kc.case_gemm_row_perm(cx, X=8, Y=48, P=96, N=320, K=320)   # output projection folding the way back
    kc.case_gemm_row_perm(cx, X=48, Y=8, P=768, N=960, K=320, form="ln")        # config #4 per-rank level 0: QKV, LayerNorm fold
    kc.case_gemm_row_perm(cx, X=8, Y=6, P=768, N=320, K=320, form="res_stats")   # ... output projection + LayerNorm statistics
    kc.case_gemm_row_perm(cx, X=6, Y=8, P=24, N=1280, K=1280, form="res_stats")  # level 2 at 8 ranks (P = 192 / 8)
    kc.case_gemm_row_perm(cx, X=6, Y=8, P=12, N=1280, K=1280)                      # level 3 (P < 16): the register-staged kernel


def test_gemm_fused(cx):
    kc.case_gemm_prologue(cx, n_img=6, rows=384, N=1280, K=1280)
    kc.case_gemm_lnfold(cx, B=2, Fr=24, P=96, C=1280, N=3840)
    kc.case_gemm_geglu(cx, M=4096, C=320)
    kc.case_geglu_pointwise(cx, M=8192, C=320)  # the LDS-DMA kernel's GEGLU epilogue
    kc.case_geglu_pointwise(cx, M=512, C=64)    # the register-staged kernel's


def test_gemm_wide_tile_kernel(cx):
    """the 256 x 320 x 64 wide-tile kernel (default for N = 320, K >= 640, M % 256 == 0) at the level-0 feed-forward output
    shape of a frame shard (M = 36 864) and with several tiles per workgroup (M = 147 456); a problem that wants statistics
    stays on the square tiles (64-row parts)"""
    for form in ("res", "ln", "plain"):
        kc.case_gemm_forms(cx, M=36864, C=1280, N=320, P=768, form=form, seed=81)
    kc.case_gemm_forms(cx, M=147456, C=640, N=320, P=6144, form="res", seed=82)
    kc.case_gn_parts_gemm(cx, n=6, rows=6144, C=320, K=1280, seed=83)
    kc.case_ln_parts_gemm(cx, M=36864, C=320, K=640, seed=84)


def test_gemm_epilogue_forms(cx):
    """the epilogue forms the engine launches (hv_gemm_epilogue_fast on the LDS-DMA kernel), every LDS-DMA tile shape"""
    for form in ("ln", "ln_yt", "ln_geglu", "res", "plain"):
        kc.case_gemm_forms(cx, M=3000, C=320, N=960, P=384, form=form)
    kc.case_gemm_forms(cx, M=4608, C=1280, N=1280, P=96, form="res", seed=35)   # level 3: 96-token images, general epilogue
    kc.case_gemm_forms(cx, M=4608, C=1280, N=3840, P=96, form="ln", seed=36)
    for variant in (0, 2, 3):
        cx.lib.call("hv_set_tuning", 3, variant)
        try:
            for form in ("ln", "ln_yt", "ln_geglu", "res", "plain"):
                kc.case_gemm_forms(cx, M=2100, C=640, N=1920, P=128, form=form, seed=37)
        finally:
            cx.lib.call("hv_set_tuning", 3, 1)


def test_bench_shape_gemm_forms(cx):
    M = 48 * 6144
    kc.case_gemm_forms(cx, M=M, C=320, N=960, P=6144, form="ln_yt")        # level-0 spatial QKV
    kc.case_gemm_forms(cx, M=M, C=320, N=960, P=6144, form="ln")           # level-0 motion-module QKV (PE row per frame)
    kc.case_gemm_forms(cx, M=M, C=320, N=1280, P=6144, form="ln_geglu")    # level-0 feed-forward input projection
    kc.case_gemm_forms(cx, M=M, C=320, N=320, P=24 * 6144, form="res")     # level-0 attention output projection (in place)
    kc.case_gemm_forms(cx, M=M, C=1280, N=320, P=24 * 6144, form="res")    # level-0 feed-forward output projection
    kc.case_gemm_forms(cx, M=48 * 1536, C=640, N=2560, P=1536, form="ln_geglu", seed=38)
    # level 2 (M = 18432): 256x256 tiles fill 1.4 / 4.2 rounds of the 256 CUs; the default takes the 128x128x64 kernel there
    for policy in (1, 2):
        cx.lib.call("hv_set_tuning", 3, policy)
        try:
            kc.case_gemm_forms(cx, M=48 * 384, C=1280, N=1280, P=24 * 384, form="res", seed=39)
            kc.case_gemm_forms(cx, M=48 * 384, C=5120, N=1280, P=24 * 384, form="res", seed=40)
            kc.case_gemm_forms(cx, M=48 * 384, C=1280, N=3840, P=384, form="ln_yt", seed=41)
        finally:
            cx.lib.call("hv_set_tuning", 3, 1)


def test_gemm_four_wave_kernel_bench_shapes(cx):
    """hv_gemm_w4_kernel (hv_gemm4.h) at the step's shapes: the default selection (deferred-store forms at K >= 1280, M >= 16384), the
    same kernel forced at level 0 (tuning 10 = 3: unit raster, five k-tiles per tile; 4: tile raster) and with its stores at
    once (2) give the bits of the 8-wave kernel (0) -- every selection accumulates k-slices in the same order."""
    import torch

    cases = [dict(M=48 * 1536, C=640, N=2560, P=1536, form="ln_geglu", seed=38),     # level-1 ff1
             dict(M=48 * 1536, C=640, N=1920, P=1536, form="ln", seed=43),           # level-1 motion-module QKV (PE row per frame)
             dict(M=48 * 384, C=1280, N=5120, P=384, form="ln_geglu", seed=44),      # level-2 ff1
             dict(M=48 * 384, C=5120, N=1280, P=24 * 384, form="res", seed=47),      # level-2 ff2: deferred residual form, in place
             dict(M=48 * 384, C=1280, N=3840, P=384, form="ln", seed=48),            # level-2 motion-module QKV (192-row tiles fill)
             dict(M=48 * 6144, C=320, N=960, P=6144, form="ln", seed=45),            # level 0 (default: 8-wave kernel) [5]
             dict(M=48 * 6144, C=320, N=1280, P=6144, form="ln_geglu", seed=46)]
    outs = {}
    try:
        for w4 in (0, 1, 3, 4, 2):
            cx.lib.call("hv_set_tuning", 10, w4)
            for i, c in enumerate(cases):
                if w4 in (4, 2) and i not in (1, 5):
                    continue
                y = kc.case_gemm_forms(cx, return_output=True, **c)
                if w4 == 0:
                    outs[i] = y
                else:
                    assert torch.equal(y, outs[i]), f"four-wave kernel (tuning 10 = {w4}) differs from the 8-wave kernel: {c}"
        # the X-stationary kernel (hv_gemm_xs.h, tuning key 11) on the two level-0 shapes
        cx.lib.call("hv_set_tuning", 10, 1)
        cx.lib.call("hv_set_tuning", 11, 1)
        for i in (5, 6):
            y = kc.case_gemm_forms(cx, return_output=True, **cases[i])
            assert torch.equal(y, outs[i]), f"X-stationary kernel differs from the 8-wave kernel: {cases[i]}"
    finally:
        cx.lib.call("hv_set_tuning", 11, 0)
        cx.lib.call("hv_set_tuning", 10, 1)


def test_gemm_c4_kernel_bench_shapes(cx):
    """hv_gemm_c4_kernel (hv_gemm_c4.h: 192 x 320 x 64 tiles on four waves) at the shapes the denoising path takes it for -- the
    feed-forward output projections of levels 0 and 1, residual in place -- and forced (tuning 13 = 2) at a short reduction:
    the bits of the default selection without it (13 = 0)"""
    import torch

    cases = [dict(M=48 * 6144, C=1280, N=320, P=6144, form="res", seed=111),    # level-0 ff2
             dict(M=48 * 1536, C=2560, N=640, P=1536, form="res", seed=112),    # level-1 ff2
             dict(M=48 * 6144, C=320, N=320, P=6144, form="res", seed=113),     # forced: five k-tiles
             dict(M=48 * 384, C=1280, N=1280, P=384, form="plain", seed=114)]   # forced: four column tiles, bias only
    try:
        for i, c in enumerate(cases):
            cx.lib.call("hv_set_tuning", 13, 0)
            ref = kc.case_gemm_forms(cx, return_output=True, res_rowvec=False, **c)
            cx.lib.call("hv_set_tuning", 13, 1 if i < 2 else 2)
            y = kc.case_gemm_forms(cx, return_output=True, res_rowvec=False, **c)
            assert torch.equal(y, ref), f"hv_gemm_c4_kernel differs from the selection without it: {c}"
    finally:
        cx.lib.call("hv_set_tuning", 13, 1)


def test_gemm_weights_in_registers_kernel(cx):
    """hv_gemm_wr_kernel (hv_gemm_wr.h) at the level-0 output projections (M = 294 912, N = K = 320): Y bit for bit as the tile
    kernels (tuning 15 = 0) with the residual in place and with bias only; LayerNorm parts (4 per row) and GroupNorm parts (one
    per 64 rows) reproduce the statistics of the stored output; a small M on one workgroup per XCD (the rings wrap 48 times)"""
    import torch

    try:
        for c in (dict(M=48 * 6144, C=320, N=320, P=6144, form="res", seed=131, res_rowvec=False),
                  dict(M=48 * 6144, C=320, N=320, P=6144, form="plain", seed=132),
                  dict(M=48 * 6144, C=320, N=320, P=24 * 6144, form="res", seed=138),      # + the folded cross-attention row per batch
                  dict(M=48 * 6144, C=320, N=960, P=6144, form="ln", seed=139)):           # motion-module QKV: LayerNorm fold + PE row per frame
            cx.lib.call("hv_set_tuning", 15, 0)
            ref = kc.case_gemm_forms(cx, return_output=True, **c)
            cx.lib.call("hv_set_tuning", 15, 1)
            y = kc.case_gemm_forms(cx, return_output=True, **c)
            assert torch.equal(y, ref), f"hv_gemm_wr_kernel differs from the tile kernels: {c}"
        kc.case_ln_parts_gemm(cx, M=48 * 6144, C=320, K=320, seed=133)
        kc.case_ln_parts_gemm(cx, M=24 * 6144, C=320, K=320, seed=134, residual=False)
        kc.case_gn_parts_gemm(cx, n=48, rows=6144, C=320, K=320, seed=135)
        cx.lib.call("hv_set_tuning", 15, 2)
        cx.lib.call("hv_set_tuning", 2, 8)
        kc.case_ln_parts_gemm(cx, M=64 * 8 * 24, C=320, K=320, seed=136)
        kc.case_gn_parts_gemm(cx, n=6, rows=1536, C=320, K=320, seed=137)
    finally:
        cx.lib.call("hv_set_tuning", 2, 512)
        cx.lib.call("hv_set_tuning", 15, 1)


def test_affine_apply(cx):
    kc.case_affine_apply(cx, n_img=48, rows=1536, C=640)
    kc.case_affine_apply(cx, n_img=48, rows=6144, C=320, act=A.ACT_SILU, seed=42)
    kc.case_affine_apply_cat(cx, n_img=48, rows=1536, C1=640, C2=320)


@pytest.mark.parametrize("mode", [A.CONV_S1, A.CONV_S2, A.CONV_UP2])
def test_conv(cx, mode):
    kc.case_conv(cx, n=4, H=48, W=32, C1=320, Cout=320, mode=mode)


def test_conv_shapes(cx):
    kc.case_conv(cx, n=3, H=24, W=16, C1=1280, C2=640, Cout=1280)
    kc.case_conv(cx, n=6, H=12, W=8, C1=1280, Cout=1280)
    kc.case_conv(cx, n=2, H=96, W=64, C1=320, Cout=4, temb=False, residual=False)
    kc.case_conv(cx, n=2, H=64, W=48, C1=32, Cout=16, pro=False, temb=False, residual=False, out_act=A.ACT_SILU)


def test_conv_four_wave_kernel(cx):
    """hv_conv_w4_kernel at the shapes the denoising path takes it for (48 images; the CPU reference on three of them: first,
    an inner one, last) and at forced shapes (tuning 12 = 2): ragged patches, two channel tiles on both rasters, GroupNorm
    partial statistics of the output"""
    kc.case_conv(cx, n=48, H=96, W=64, C1=320, Cout=320, pro=False, check=(0, 17, 47), seed=81)       # level-0 conv2 (residual)
    kc.case_conv(cx, n=48, H=48, W=32, C1=640, Cout=640, pro=False, check=(0, 47), seed=82)           # level 1, raster 1
    kc.case_conv(cx, n=48, H=96, W=64, C1=640, Cout=320, pro=False, residual=False, check=(0, 47), seed=83)  # up path, 10 chunks
    kc.case_conv(cx, n=48, H=48, W=32, C1=640, Cout=640, mode=A.CONV_UP2, pro=False, temb=False, residual=False, check=(0, 47), seed=88)  # upsample 1 -> 0
    kc.case_conv(cx, n=48, H=12, W=8, C1=1280, Cout=1280, mode=A.CONV_UP2, pro=False, temb=False, residual=False, check=(0, 47), seed=89)  # upsample 3 -> 2
    cx.lib.call("hv_set_tuning", 12, 2)
    try:
        kc.case_conv(cx, n=2, H=15, W=9, C1=128, Cout=320, mode=A.CONV_UP2, pro=False, seed=90)       # upsample-folded, ragged
        kc.case_gn_parts_conv(cx, n=2, H=24, W=16, Cin=64, Cout=320, mode=A.CONV_UP2, seed=58)
        kc.case_conv(cx, n=3, H=64, W=64, C1=320, Cout=320, pro=False, seed=84)                       # config #2: ragged rows
        kc.case_conv(cx, n=2, H=30, W=72, C1=128, Cout=640, pro=False, out_act=A.ACT_SILU, seed=85)   # ragged both ways
        kc.case_conv(cx, n=2, H=24, W=16, C1=1280, Cout=1280, pro=False, seed=86)                     # level 2: 20 chunks, 4 channel tiles
        cx.lib.call("hv_set_tuning", 9, 0)
        kc.case_conv(cx, n=2, H=24, W=16, C1=640, Cout=640, pro=False, seed=87)                       # raster 0
        cx.lib.call("hv_set_tuning", 9, 2)
        kc.case_gn_parts_conv(cx, n=2, H=16, W=16, Cin=64, Cout=320)
        kc.case_gn_parts_conv(cx, n=3, H=48, W=32, Cin=128, Cout=640, offset=3.0, seed=56)
        # same reduction order as hv_conv3x3_kernel's 64-channel-chunk variant: the same bits (level 2 at 48 images)
        import torch
        outs = {}
        for name, (w4, big) in dict(w4=(2, 1), ck64=(0, 3)).items():
            cx.lib.call("hv_set_tuning", 12, w4)
            cx.lib.call("hv_set_tuning", 5, big)
            outs[name] = kc.case_conv(cx, n=48, H=24, W=16, C1=1280, Cout=1280, pro=False, check=(0,), seed=91, return_output=True)
        assert torch.equal(outs["w4"], outs["ck64"]), "hv_conv_w4_kernel differs from the 64-channel-chunk kernel"
    finally:
        cx.lib.call("hv_set_tuning", 5, 1)
        cx.lib.call("hv_set_tuning", 9, 2)
        cx.lib.call("hv_set_tuning", 12, 1)


def test_layernorm_stats(cx):
    for C, M in ((320, 48 * 6144), (640, 48 * 1536 + 3), (1280, 4608), (192, 1000)):
        kc.case_layernorm_stats(cx, M=M, C=C)


def test_groupnorm(cx):
    kc.case_groupnorm(cx, n=4, H=96, W=64, C1=320)
    kc.case_groupnorm(cx, n=4, H=24, W=16, C1=1280, C2=640)
    kc.case_groupnorm(cx, n=3, H=12, W=8, C1=2560)
    kc.case_groupnorm(cx, n=2, H=96, W=64, C1=320, offset=40.0, spread=0.05, splits=64)  # |mean| >> std (Chan / Welford merge)
    kc.case_groupnorm(cx, n=2, H=24, W=16, C1=1280, C2=640, offset=-25.0, spread=0.2, splits=8)
    kc.case_groupnorm(cx, n=2, H=3, W=3, C1=640, seed=14, splits=4)  # an empty last pixel range


@pytest.mark.parametrize("D,Lq,Lb", [(40, 1536, 1536), (40, 200, 72), (80, 384, 384), (160, 96, 96), (160, 384, 96), (40, 144, 144),
                                     (80, 72, 200)])
def test_attention(cx, D, Lq, Lb):
    kc.case_attention(cx, D=D, n_img=4, Lq=Lq, Lb=Lb)


@pytest.mark.parametrize("D,L", [(40, 1536), (80, 768), (160, 384)])
def test_attention_fp8(cx, D, L):
    """e4m3 QK^T / PV (hv_attention_fp8): i.i.d. random operands, bound 6.6e-2 (see case_attention), forced maximum jumps"""
    kc.case_attention(cx, D=D, n_img=4, Lq=L, Lb=L, fp8=True, seed=75, check=(0, 3), q_stride=4)
    kc.case_attention(cx, D=D, n_img=4, Lq=L + 40, Lb=L - 24, fp8=True, spike=True, seed=76, check=(1, 2), q_stride=4)


def test_bench_shape_attention_fp8(cx):
    kc.case_attention(cx, D=40, n_img=48, Lq=6144, Lb=6144, fp8=True, check=(0, 47), q_stride=16)


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("D,L", [(40, 9216), (80, 2304), (160, 576), (160, 144)])
def test_config5_shape_attention(cx, D, L, fp8):
    """the spatial-attention geometries of BASELINE.json configs[4] (48f x 1024x576: latent 128 x 72 -> 9216 / 2304 / 576 /
    144 tokens per image at the four levels), bf16 and fp8 kernels, 48 images of one 24-frame window with CFG"""
    kc.case_attention(cx, D=D, n_img=48, Lq=L, Lb=L, fp8=fp8, seed=77, check=(0, 47), q_stride=16 if L > 1000 else 1)


def test_attention_variants(cx):
    """head dim 40: the generic kernel (tuning value 2) computes the same function as the dedicated one (0)"""
    try:
        for v in (2, 0):
            cx.lib.call("hv_set_tuning", 0, v)
            kc.case_attention(cx, D=40, n_img=4, Lq=520, Lb=264)
            kc.case_attention(cx, D=40, n_img=4, Lq=1536, Lb=1536, spike=True, seed=73, check=(0, 3), q_stride=4)
    finally:
        cx.lib.call("hv_set_tuning", 0, 0)
    kc.case_attention(cx, D=160, n_img=4, Lq=96, Lb=96)


@pytest.mark.parametrize("D,L", [(40, 1536), (80, 768), (160, 384)])
def test_attention_forced_rescale(cx, D, L):
    kc.case_attention(cx, D=D, n_img=4, Lq=L, Lb=L, spike=True, seed=73, check=(0, 3), q_stride=4)
    kc.case_attention(cx, D=D, n_img=4, Lq=L + 40, Lb=L - 24, spike=True, seed=74, check=(1, 2), q_stride=4)
    if D == 40:  # exp2 overflow against the first tile's reference: the optimistic kernel's careful second pass
        kc.case_attention(cx, D=D, n_img=4, Lq=L, Lb=L, spike=8.0, seed=75, check=(0, 3), q_stride=4)
        kc.case_attention(cx, D=D, n_img=4, Lq=L + 40, Lb=L - 24, spike=8.0, seed=76, check=(1, 2), q_stride=4)


@pytest.mark.parametrize("D,Fr,P", [(40, 24, 384), (80, 16, 96), (160, 24, 24), (40, 8, 64), (80, 32, 40), (40, 18, 50), (80, 24, 768),
                                     (160, 24, 384)])
def test_temporal(cx, D, Fr, P):
    kc.case_temporal(cx, D=D, B=2, Fr=Fr, P=P)




# ---- the exact shapes of the config-#3 benchmark step (48 images of 96x64 latents): grid-size dependent faults only show
# here.  The kernels run the full problem; the CPU reference is evaluated in full where it is cheap and on a subset of
# images / query rows where it is not (conv, attention).
def test_bench_shape_gemms(cx):
    M = 48 * 6144
    kc.case_gemm(cx, M=M, N=960, K=320, transposed=True)                # level-0 QKV (V^T tail)
    kc.case_gemm(cx, M=M, N=320, K=1280)                                # level-0 ff2 / out-projection family
    kc.case_gemm_geglu(cx, M=M, C=320)                                  # level-0 ff1 + GEGLU
    kc.case_gemm_lnfold(cx, B=2, Fr=24, P=6144, C=320, N=960)           # motion-module QKV (LN + PE folded)
    kc.case_gemm_prologue(cx, n_img=48, rows=6144, N=320, K=320)        # proj_in with the GroupNorm prologue
    kc.case_gemm(cx, M=48 * 96, N=1280, K=5120)                         # level-3 ff2


def test_bench_shape_convs(cx):
    kc.case_conv(cx, n=48, H=96, W=64, C1=320, Cout=320, check=(0, 31, 47))
    kc.case_conv(cx, n=48, H=96, W=64, C1=320, C2=320, Cout=320, check=(5, 46))          # up-block concat
    kc.case_conv(cx, n=48, H=48, W=32, C1=640, Cout=640, mode=A.CONV_UP2, temb=False, residual=False, pro=False,
                 check=(0, 47))                                                           # Upsample3D into level 0
    kc.case_conv(cx, n=48, H=96, W=64, C1=320, Cout=320, mode=A.CONV_S2, temb=False, residual=False, pro=False,
                 check=(1, 40))
    kc.case_conv(cx, n=48, H=12, W=8, C1=1280, Cout=1280, check=(0, 24, 47))


@pytest.mark.parametrize("D,L", [(40, 6144), (80, 1536), (160, 384), (160, 96)])
def test_bench_shape_attention(cx, D, L):
    kc.case_attention(cx, D=D, n_img=48, Lq=L, Lb=L, check=(0, 23, 24, 47), q_stride=8 if L > 1000 else 1,
                      )


@pytest.mark.parametrize("D,P", [(40, 6144), (80, 1536), (160, 384), (160, 96)])
def test_bench_shape_temporal(cx, D, P):
    kc.case_temporal(cx, D=D, B=2, Fr=24, P=P)


def test_elementwise(cx):
    kc.case_elementwise(cx)


def test_errors_are_python_exceptions(cx):
    x = torch.zeros(8, 100, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(16, 100, dtype=torch.bfloat16, device="cuda")
    y = torch.zeros(8, 16, dtype=torch.bfloat16, device="cuda")
    from humanvid_amd import ops

    with pytest.raises(ValueError):
        ops.gemm(cx.lib, cx.stream, x, w, y)  # K % 64 != 0
    with pytest.raises(NotImplementedError):
        ops.attention(cx.lib, cx.stream, x, x, x, y, n_images=1, heads=8, D=64, Lq=8, L1=8, ldq=512, ldk=512,
                      ldvt=8, ldo=512)


def test_groupnorm_statistics_from_the_producing_kernels(cx):
    """hv_conv3x3 / hv_gemm gn_part + hv_groupnorm_from_parts against the statistics pass and torch, incl. bench geometries"""
    kc.case_gn_parts_conv(cx, n=2, H=16, W=16, Cin=32, Cout=320)
    kc.case_gn_parts_conv(cx, n=2, H=12, W=8, Cin=64, Cout=1280, offset=3.0)            # level 3: narrow, ragged rows, 64-channel chunks
    kc.case_gn_parts_conv(cx, n=3, H=24, W=16, Cin=128, Cout=640, C2=320, seed=53)      # up-block concat, groups of 30 straddle the seam
    kc.case_gn_parts_conv(cx, n=2, H=24, W=16, Cin=64, Cout=128, mode=A.CONV_UP2, seed=54)   # 256-pixel patches, four pixel quarters
    kc.case_gn_parts_conv(cx, n=2, H=48, W=32, Cin=64, Cout=64, mode=A.CONV_S2, seed=55)
    kc.case_gn_parts_conv(cx, n=4, H=96, W=64, Cin=32, Cout=320, seed=56, offset=2.0)   # level 0
    kc.case_gn_parts_gemm(cx, n=3, rows=128, C=320, K=64)
    kc.case_gn_parts_gemm(cx, n=6, rows=1536, C=640, K=640, seed=57)                    # level 1 projection-out
    kc.case_gn_parts_gemm(cx, n=4, rows=6144, C=320, K=320, seed=58, offset=2.0)        # level 0
    kc.case_ln_parts_gemm(cx, M=300, C=128, K=64)
    kc.case_ln_parts_gemm(cx, M=48 * 1536, C=640, K=640, seed=60, offset=1.5)                # level 1 attention out-projection
    kc.case_ln_parts_gemm(cx, M=24 * 6144, C=320, K=320, residual=False, seed=61)           # level 0 proj_in
