#!/usr/bin/env python3
"""Pin the ReferenceNet write pass against the REFERENCE's own UNet2DConditionModel  --  TEST INFRASTRUCTURE ONLY.

    python oracle/gen_refnet_golden.py [--check]

Build container only (needs /root/reference).  Imports /root/reference/src/models/unet_2d_condition.py (+ its
unet_2d_blocks.py, transformer_2d.py, attention.py, mutual_self_attention.py) verbatim on top of oracle/refshim --
which, for this, also restates diffusers 0.24's ResnetBlock2D / Downsample2D / Upsample2D / LoRACompatible* --,
builds the SD-1.5 geometry, loads oracle_torch.make_reference_net_weights() with strict=True (state-dict grammar),
installs ReferenceAttentionControl(mode="write", fusion_blocks="full") and runs the write pass of
pipeline_pose2vid_long.py:470-480 (ref latent repeated for the zero-CLIP and the CLIP entry, t = 0).  Checks:

  * the 16 banks equal oracle_torch.reference_net_banks (the restatement the GPU tests use)       -> pins the oracle
  * bank entry 1 is unchanged when the pass is run on the conditional entry alone (batch 1)        -> SURVEY.md 8f-1
and writes tests/golden/refnet_sd15.npz (inputs + banks, fp16) for tests/test_gpu_refnet.py.
"""
import argparse
import contextlib
import io
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refenv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    if not refenv.present():
        print("reference tree not present; nothing to do")
        return 2
    shim, O = refenv.enter()
    import numpy as np
    import torch

    torch.set_grad_enabled(False)
    from src.models.attention import BasicTransformerBlock
    from src.models.mutual_self_attention import ReferenceAttentionControl
    from src.models.unet_2d_condition import UNet2DConditionModel

    cfg = dict(O.SD15_UNET3D_CFG)
    sd = O.make_reference_net_weights(cfg, seed=5)
    sd15 = dict(  # stable-diffusion-v1-5/unet/config.json
        sample_size=64, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
        down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
        up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
        block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1, mid_block_scale_factor=1,
        act_fn="silu", norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=768, attention_head_dim=8)
    with contextlib.redirect_stdout(io.StringIO()):
        ref = UNet2DConditionModel(**sd15).eval()
    missing, unexpected = ref.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    print(f"reference UNet2DConditionModel (ReferenceNet): {len(sd)} tensors, strict load ok")

    hh, ww = 32, 16
    g = torch.Generator().manual_seed(8)
    lat = torch.randn(1, 4, hh, ww, generator=g)
    clip = torch.randn(1, 1, 768, generator=g)
    ehs = torch.cat([torch.zeros_like(clip), clip])
    writer = ReferenceAttentionControl(ref, do_classifier_free_guidance=True, mode="write", batch_size=1,
                                       fusion_blocks="full")
    with contextlib.redirect_stdout(io.StringIO()):
        ref(lat.repeat(2, 1, 1, 1), torch.zeros(()), encoder_hidden_states=ehs, return_dict=False)
    blocks = {n.rsplit(".transformer_blocks.0", 1)[0]: m for n, m in ref.named_modules()
              if isinstance(m, BasicTransformerBlock)}
    banks = {loc: m.bank[0].clone() for loc, m in blocks.items()}
    assert len(banks) == 16 and all(len(m.bank) == 1 for m in blocks.values())
    ora = O.reference_net_banks(sd, cfg, lat.repeat(2, 1, 1, 1), ehs)
    assert set(ora) == set(banks)
    worst = 0.0
    for loc, want in banks.items():
        err = float((ora[loc] - want).abs().max())
        worst = max(worst, err)
        assert err <= 2e-5 * max(1.0, float(want.abs().max())), (loc, err)
    print(f"  ok oracle reference_net_banks == reference write pass: max|diff| over 16 banks {worst:.3e}")
    # the conditional entry alone
    writer.clear()
    with contextlib.redirect_stdout(io.StringIO()):
        ref(lat, torch.zeros(()), encoder_hidden_states=ehs[1:], return_dict=False)
    w1 = 0.0
    for loc, m in blocks.items():
        w1 = max(w1, float((m.bank[0][0] - banks[loc][1]).abs().max()))
    assert w1 <= 1e-5, w1
    print(f"  ok batch-1 write pass reproduces bank entry 1: max|diff| {w1:.3e}")
    if not args.check:
        out = os.path.join(refenv.REPO, "tests", "golden", "refnet_sd15.npz")
        np.savez_compressed(out, lat=lat.numpy(), clip=clip.numpy(),
                            **{"bank:" + k: v.numpy().astype(np.float16) for k, v in banks.items()})
        print("wrote", out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
