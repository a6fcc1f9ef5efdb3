"""Parity checks that can see a kernel regression.

The full-size comparison with the REFERENCE's own fp32 forward (tests/test_gpu_fullsize_parity.py) has to allow 2e-2, because
the bf16 storage format of weights and activations alone accounts for ~1.3e-2 of the measured 1.44e-2: a kernel that added
5e-3 of its own would still pass.  The comparator here is oracle/storage_model.py: the reference-pinned fp32 oracle WITH the
native path's storage points (bf16 roundings of the weights, of every kernel-boundary activation, of the residual stream, of
the re-scaled query operand, of the softmax probabilities against the kernels' reference maximum, of the GEGLU hidden state;
pinned to oracle_torch by tests/test_storage_model.py).

Two checks, both at config #3's full size (24 frames, latent 96x64, SD-1.5 widths, CFG, banks):

1. TEACHER-FORCED BLOCKS.  bf16 rounding noise is chaotic -- a perturbation of 1e-6 flips a rounding somewhere, the flipped
   rounding is a full-size perturbation of everything downstream -- so two faithful implementations of the same storage model
   decorrelate with depth (check 2 measures it).  A regression detector therefore has to compare ONE block at a time from
   identical inputs: the native path's own stored activation in front of a resnet / spatial transformer / motion module is fed
   to the storage model's block, and its output is compared with what the native path stored behind that block (the two
   largest transformer blocks: on 12 of the 48 images, both CFG halves -- FRAME_SUBSET).  26 blocks: ten on the down path,
   and since round 6 the mid block (transformer at 96 tokens / head dim 160, motion module, resnet), a stride-2 downsample, both
   kinds of upsample-folded convolution, three up-path resnets with the two-source (concatenated) GroupNorm + convolution +
   1x1 shortcut GEMM at levels 2 / 1 / 0, an up-path transformer and motion module, and conv_norm_out + conv_out.  They
   cover every kernel family at its benchmarked shapes (3x3 convolutions at 320 / 1280 / 1280 channels, the GEMM epilogue
   forms, spatial attention at head dims 40 / 80 / 160 with bank keys and the CFG halves, temporal attention, GroupNorm /
   LayerNorm statistics from the producers' partial sums).  Stated bounds: resnet <= 2e-3, spatial transformer <= 3.5e-3,
   motion module <= 4e-3 (measured on MI355X: 5.0e-4 .. 9.6e-4, 1.7e-3 .. 2.0e-3, 2.1e-3 .. 2.9e-3 -- DESIGN.md section 4;
   what is left is the decorrelation of a block's own 4 - 12 storage points, about one bf16 rounding rms each) -- a kernel
   that adds 5e-3 to a block fails it.
2. END TO END against tests/golden/unet3d_config3_storage.npz (oracle/gen_storage_model_golden.py ran the storage model once
   at this size): the per-image rms of every tap agrees within 5e-4 for all 48 images (against the reference: 2 %), while
   the element-wise distance grows from 1.3e-3 behind the first resnet to ~1.2e-2 at the end -- two realisations of the same
   rounding noise, not an error of either; bound 2e-2 as against the reference.
"""
import os
import sys
import time

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
sys.path.insert(0, os.path.dirname(__file__))
import fullsize_case as FC  # noqa: E402
import oracle_torch as O  # noqa: E402  (test infrastructure: weight generator only)
import storage_model as SM  # noqa: E402  (test infrastructure: the comparator)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASE = "config3"
# (kind, weight prefix, tap in front of the block, tap behind it)
BLOCKS = [
    ("resnet", "down_blocks.0.resnets.1", "down_blocks.0.motion_modules.0", "down_blocks.0.resnets.1"),
    ("transformer", "down_blocks.0.attentions.1", "down_blocks.0.resnets.1", "down_blocks.0.attentions.1"),
    ("motion", "down_blocks.0.motion_modules.1", "down_blocks.0.attentions.1", "down_blocks.0.motion_modules.1"),
    ("transformer", "down_blocks.1.attentions.1", "down_blocks.1.resnets.1", "down_blocks.1.attentions.1"),
    ("motion", "down_blocks.1.motion_modules.1", "down_blocks.1.attentions.1", "down_blocks.1.motion_modules.1"),
    ("resnet", "down_blocks.2.resnets.1", "down_blocks.2.motion_modules.0", "down_blocks.2.resnets.1"),
    ("transformer", "down_blocks.2.attentions.1", "down_blocks.2.resnets.1", "down_blocks.2.attentions.1"),
    ("motion", "down_blocks.2.motion_modules.1", "down_blocks.2.attentions.1", "down_blocks.2.motion_modules.1"),
    ("resnet", "down_blocks.3.resnets.1", "down_blocks.3.motion_modules.0", "down_blocks.3.resnets.1"),
    ("motion", "down_blocks.3.motion_modules.1", "down_blocks.3.resnets.1", "down_blocks.3.motion_modules.1"),
    # round 6 (the down path was a third of the network): the mid block, the up path with its two-source
    # (concatenated) convolutions, the resampling convolutions, conv_out -- "@name" = a fine tap (engine.tap_fine)
    ("transformer", "mid_block.attentions.0", "@mid_block.resnets.0", "@mid_block.attentions.0"),
    ("motion", "mid_block.motion_modules.0", "@mid_block.attentions.0", "@mid_block.motion_modules.0"),
    ("resnet", "mid_block.resnets.1", "@mid_block.motion_modules.0", "mid_block"),
    ("down", "down_blocks.1.downsamplers.0", "down_blocks.1.motion_modules.1", "@down_blocks.1.downsamplers.0"),
    ("up", "up_blocks.0.upsamplers.0", "up_blocks.0.2", "@up_blocks.0.upsamplers.0"),       # 1280 ch, 12x8 -> 24x16: the upsample-folded kernel
    ("up", "up_blocks.2.upsamplers.0", "up_blocks.2.2", "@up_blocks.2.upsamplers.0"),       # 640 ch, 48x32 -> 96x64
    ("resnet2", "up_blocks.1.resnets.0", "@up_blocks.0.upsamplers.0", "@up_blocks.1.resnets.0", "down_blocks.2.motion_modules.1"),
    ("transformer", "up_blocks.1.attentions.0", "@up_blocks.1.resnets.0", "@up_blocks.1.attentions.0"),
    ("resnet2", "up_blocks.2.resnets.2", "up_blocks.2.1", "@up_blocks.2.resnets.2", "@down_blocks.0.downsamplers.0"),
    ("resnet2", "up_blocks.3.resnets.2", "up_blocks.3.1", "@up_blocks.3.resnets.2", "@conv_in"),
    ("motion", "up_blocks.3.motion_modules.2", "@up_blocks.3.attentions.2", "up_blocks.3.2"),
    ("conv_out", "conv_out", "up_blocks.3.2", "@conv_out"),
]
TOL_BLOCK = dict(resnet=2e-3, resnet2=2e-3, transformer=3.5e-3, motion=4e-3, down=5e-4, up=5e-4, conv_out=2e-3)
# A spatial transformer block is independent per image (GroupNorm, attention and LayerNorm all act inside one image; the bank
# keys and the constant cross-attention term belong to the CFG half), and the storage model's fp32 attention on the host is what
# this file spends its time on (level 0: 131 s for the 48 images): the two largest blocks are compared on an evenly spaced
# subset of the frames of BOTH halves -- every image of the native forward ran, a subset is checked (the driver's GPU tier has a
# 20-minute budget for the whole suite).
FRAME_SUBSET = {"down_blocks.0.attentions.1": 12, "down_blocks.1.attentions.1": 12}
TOL_E2E, TOL_RMS = 2e-2, 5e-4


@pytest.fixture(scope="module")
def run():
    """one native forward at config #3; keeps the full activations the teacher-forced blocks need (fp32 copies on the host)"""
    from humanvid_amd.unet3d import UNet3DConditionModel

    cfg = dict(O.SD15_UNET3D_CFG)
    sd = O.make_unet3d_weights(cfg, seed=FC.WEIGHT_SEED)
    kw = dict(cfg, use_inflated_groupnorm=True, unet_use_cross_frame_attention=False, unet_use_temporal_attention=False,
              motion_module_type="Vanilla")
    net = UNet3DConditionModel(**kw)
    net.load_state_dict(sd, strict=True)
    chan = {p: sd[p + ".norm.weight"].numel() for p in O.transformer_locations(cfg)}
    net = net.to("cuda")
    F = FC.CASES[CASE]["F"]
    sample, ehs, pose, banks = FC.make_inputs(CASE, list(chan), lambda p: chan[p])
    eng = net.engine()
    eng.set_reference_banks({k: v.cuda() for k, v in banks.items()}, do_cfg=True)
    eng._banks_from_modules = lambda: None
    keep = {t for b in BLOCKS for t in b[2:]}
    full, got_slice, got_rms = {}, {}, {}

    def tap(name, x):  # x: [(b f), h, w, c] bf16, possibly a buffer that a later block updates in place
        got_slice[name] = FC.slice_nhwc(x, F).float().cpu()
        got_rms[name] = FC.rms_nhwc(x).cpu()
        if name in keep:
            full[name] = x.float().permute(0, 3, 1, 2).contiguous().cpu()  # -> (b f) c h w, what the oracle blocks take

    def tap_fine(name, x):  # the activations between the 35 tap points (only what a block of BLOCKS needs is kept)
        if "@" + name in keep:
            full["@" + name] = x.float().permute(0, 3, 1, 2).contiguous().cpu()

    eng.tap, eng.tap_fine = tap, tap_fine
    try:
        out = net(sample.cuda(), FC.TIMESTEP, ehs.cuda(), pose_cond_fea=pose.cuda(), return_dict=False)[0]
        torch.cuda.synchronize()
    finally:
        eng.tap = eng.tap_fine = None
    assert torch.isfinite(out).all()
    return dict(cfg=cfg, sd=sd, F=F, ehs=ehs, banks=banks, out=out.float().cpu(), full=full, slices=got_slice, rms=got_rms)


def test_teacher_forced_blocks_match_the_storage_model(run):
    torch.set_grad_enabled(False)
    cfg, sd, F = run["cfg"], run["sd"], run["F"]
    model = SM.StorageModel(sd, cfg)
    temb = model.time_embedding(FC.TIMESTEP, 2, F)
    mmk = cfg["motion_module_kwargs"]
    worst = {}
    import torch.nn.functional as TF

    for kind, prefix, t_in, t_out, *t_skip in BLOCKS:
        x, want = run["full"][t_in], run["full"][t_out]
        t0 = time.time()
        if kind == "resnet":
            y = model.resnet(prefix, x, None, temb)
        elif kind == "resnet2":  # the up path: channels of the skip connection concatenated behind x (unet_3d_blocks.py:715-717)
            y = model.resnet(prefix, x, run["full"][t_skip[0]], temb)
        elif kind == "down":
            y = model.q(model.conv(prefix + ".conv", x, stride=2, padding=1))
        elif kind == "up":
            y = model.q(model.conv(prefix + ".conv", TF.interpolate(x, scale_factor=2.0, mode="nearest")))
        elif kind == "conv_out":  # conv_norm_out -> SiLU -> conv_out (the 4 of 8 padded output channels that exist)
            y = model.q(model.conv("conv_out", model.q(TF.silu(model.gn("conv_norm_out", x, model.eps)))))
            want = want[:, : y.shape[1]]
        elif kind == "transformer":
            fs = F
            if prefix in FRAME_SUBSET:
                fs = FRAME_SUBSET[prefix]
                idx = torch.linspace(0, F - 1, fs).round().long()
                pick = lambda t: t.view(2, F, *t.shape[1:])[:, idx].reshape(2 * fs, *t.shape[1:])  # noqa: E731
                x, want = pick(x), pick(want)
            y = model.transformer(prefix, x, run["ehs"], run["banks"][prefix], fs, True)
        else:
            y = model.motion(prefix, x, F, mmk)
        e = float((want - y).norm() / y.norm())
        d = (want - y).pow(2).sum(dim=(1, 2, 3)).sqrt() / y.pow(2).sum(dim=(1, 2, 3)).sqrt()  # per image
        print(f"[{CASE} teacher-forced] {prefix:34s} {kind:11s} nrmse {e:.3e}  worst image {float(d.max()):.3e}  "
              f"(storage model on the host: {time.time() - t0:.0f} s)", flush=True)
        worst[prefix] = (kind, e, float(d.max()))
    for prefix, (kind, e, dmax) in worst.items():
        assert e < TOL_BLOCK[kind] and dmax < 1.25 * TOL_BLOCK[kind], (prefix, kind, e, dmax)


def test_end_to_end_against_the_storage_model_golden(run):
    z = np.load(os.path.join(GOLD, f"unet3d_{CASE}_storage.npz"))
    ref = torch.from_numpy(z["out"].astype(np.float32))
    out = run["out"]
    e_out = float((out - ref).norm() / ref.norm())
    print(f"[{CASE} vs storage model, end to end] output nrmse {e_out:.4e}   (storage model vs the reference's fp32 forward: "
          f"{float(z['floor_out']):.4e})")
    names = [k[4:] for k in z.files if k.startswith("tap:")]
    assert len(names) == 35 and set(names) == set(run["slices"])
    worst_tap, worst_rms = ("", 0.0), ("", 0.0)
    for name in names:
        want = torch.from_numpy(z["tap:" + name].astype(np.float32))
        e = float((run["slices"][name] - want).norm() / want.norm())
        rr = torch.from_numpy(z["rms:" + name])
        er = float(((run["rms"][name] - rr).abs() / rr).max())
        print(f"[{CASE} vs storage model, end to end] {name:34s} slice nrmse {e:.3e}   per-image rms dev {er:.3e}")
        worst_tap = max(worst_tap, (name, e), key=lambda t: t[1])
        worst_rms = max(worst_rms, (name, er), key=lambda t: t[1])
    assert worst_rms[1] < TOL_RMS, worst_rms
    assert worst_tap[1] < TOL_E2E and e_out < TOL_E2E, (worst_tap, e_out)
