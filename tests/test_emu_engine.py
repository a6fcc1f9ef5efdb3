"""Launch-sequence check of the native UNet executor on the host emulator (no GPU needed).

The executor's Python schedule (humanvid_amd/engine.py) is driven with CPU tensors and the
emulated kernel library injected in place of libhumanvid_hip.so -- test-only plumbing: the product
entry points (`UNet3DConditionModel.forward`) refuse CPU tensors.  Smallest geometry that still has
every block type (resnet with/without shortcut, skip concat, down/up-sampling, spatial transformer
with bank + CFG halves, motion modules), compared with the oracle.
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "emu"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import build_emu  # noqa: E402
import oracle_torch as O  # noqa: E402

from humanvid_amd import _abi as A  # noqa: E402
from humanvid_amd import ops, packing  # noqa: E402

BF16 = torch.bfloat16


def _engine_on_emulator(net, kind="denoise"):
    from humanvid_amd import engine as E
    from humanvid_amd import lib as hvlib

    emu = A.HvLibrary(build_emu.build())
    hvlib._LIB, saved = emu, hvlib._LIB  # inject for construction; restored below
    hvlib_stream = hvlib.current_stream
    hvlib.current_stream = lambda: None
    try:
        eng = E.UNet3DEngine(net, device=torch.device("cpu"), kind=kind)
    finally:
        hvlib._LIB = saved
    eng.lib = emu
    eng.run.lib = emu
    return eng, (hvlib, hvlib_stream)


@pytest.mark.skipif(os.environ.get("HV_SLOW") != "1", reason="~5 min on the host emulator; set HV_SLOW=1")
def test_unet_schedule_on_emulator():
    from humanvid_amd.unet3d import UNet3DConditionModel

    cfg = O.tiny_unet3d_cfg()
    sd = O.make_unet3d_weights(cfg, seed=0)
    kw = dict(cfg, use_inflated_groupnorm=True, unet_use_cross_frame_attention=False,
              unet_use_temporal_attention=False, motion_module_type="Vanilla")
    net = UNet3DConditionModel(**kw)
    net.load_state_dict(sd, strict=True)
    eng, (hvlib, restore) = _engine_on_emulator(net)
    try:
        g = torch.Generator().manual_seed(42)
        b, f, hh, ww = 2, 2, 8, 4
        sample = torch.randn(1, 4, f, hh, ww, generator=g).repeat(2, 1, 1, 1, 1)
        ehs = torch.cat([torch.zeros(1, 1, 768), torch.randn(1, 1, 768, generator=g)])
        pose = (torch.randn(1, 320, f, hh, ww, generator=g) * 0.5).repeat(2, 1, 1, 1, 1)
        banks = {}
        for p in O.transformer_locations(cfg):
            c = sd[p + ".norm.weight"].numel()
            n_tok = hh * ww if c == 320 else (hh // 2) * (ww // 2)
            n_tok = max(8, n_tok)  # bank length is independent of the latent size; keep it % 8
            banks[p] = torch.randn(2, n_tok, c, generator=g).half().float()
        ref = O.unet3d_forward(sd, cfg, sample, 601, ehs, pose, banks, do_cfg=True)

        eng.set_encoder_hidden_states(ehs)
        eng.set_reference_banks(banks, do_cfg=True)
        x_in = torch.zeros(b * f, hh, ww, 32, dtype=BF16)
        ops.pack_ncfhw(eng.lib, None, sample.contiguous(), x_in)
        cond = torch.zeros(b * f, hh, ww, 320, dtype=BF16)
        ops.pack_ncfhw(eng.lib, None, pose.contiguous(), cond)
        y = eng.forward_nhwc(x_in, torch.full((b,), 601.0), cond, B=b, F=f)
        out = torch.zeros(b, 4, f, hh, ww)
        ops.unpack_nhwc(eng.lib, None, y, out)
        e = float((out - ref).norm() / ref.norm())
        print("emulated UNet forward vs oracle: nrmse", e)
        assert torch.isfinite(out).all()
        assert e < 2e-2, e
    finally:
        hvlib.current_stream = restore
