"""Geometry of the denoising UNet (block wiring, channel bookkeeping) shared by the parameter
containers (`unet3d.py`) and the native executor (`engine.py`).

Restates the constructor logic of the reference so that parameter names/shapes are
state-dict compatible: /root/reference/src/models/unet_3d.py:83-248 and
/root/reference/src/models/unet_3d_blocks.py:296-396 (CrossAttnDownBlock3D), 467-538 (DownBlock3D),
171-267 (mid), 586-680 (CrossAttnUpBlock3D), 749-814 (UpBlock3D).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple


@dataclass
class BlockSpec:
    prefix: str
    kind: str  # "down" | "mid" | "up"
    channels: int
    has_attn: bool
    has_motion: bool
    resnets: List[Tuple[int, int, int]] = field(default_factory=list)  # (c_main, c_skip, c_out)
    resample: bool = False  # downsampler (down) / upsampler (up)


def build_block_specs(cfg: dict) -> List[BlockSpec]:
    boc = tuple(cfg["block_out_channels"])
    nblk = len(boc)
    lpb = int(cfg.get("layers_per_block", 2))
    use_mm = bool(cfg.get("use_motion_module", False))
    mm_res = tuple(cfg.get("motion_module_resolutions", (1, 2, 4, 8)))
    dec_only = bool(cfg.get("motion_module_decoder_only", False))
    specs: List[BlockSpec] = []
    out_ch = boc[0]
    for i, btype in enumerate(cfg["down_block_types"]):
        if btype not in ("CrossAttnDownBlock3D", "DownBlock3D"):
            raise ValueError(f"{btype} does not exist.")
        in_ch, out_ch = out_ch, boc[i]
        specs.append(
            BlockSpec(
                prefix=f"down_blocks.{i}", kind="down", channels=out_ch,
                has_attn=btype == "CrossAttnDownBlock3D",
                has_motion=use_mm and (2**i in mm_res) and not dec_only,
                resnets=[(in_ch if j == 0 else out_ch, 0, out_ch) for j in range(lpb)],
                resample=i != nblk - 1,
            )
        )
    if cfg.get("mid_block_type", "UNetMidBlock3DCrossAttn") != "UNetMidBlock3DCrossAttn":
        raise ValueError(f"unknown mid_block_type : {cfg.get('mid_block_type')}")
    specs.append(
        BlockSpec(prefix="mid_block", kind="mid", channels=boc[-1], has_attn=True,
                  has_motion=use_mm and bool(cfg.get("motion_module_mid_block", False)),
                  resnets=[(boc[-1], 0, boc[-1]), (boc[-1], 0, boc[-1])])
    )
    rev = list(reversed(boc))
    out_ch = rev[0]
    for i, btype in enumerate(cfg["up_block_types"]):
        if btype not in ("CrossAttnUpBlock3D", "UpBlock3D"):
            raise ValueError(f"{btype} does not exist.")
        prev_out, out_ch = out_ch, rev[i]
        in_ch = rev[min(i + 1, nblk - 1)]
        nl = lpb + 1
        resnets = []
        for j in range(nl):
            skip = in_ch if j == nl - 1 else out_ch
            main = prev_out if j == 0 else out_ch
            resnets.append((main, skip, out_ch))
        specs.append(
            BlockSpec(prefix=f"up_blocks.{i}", kind="up", channels=out_ch, has_attn=btype == "CrossAttnUpBlock3D",
                      has_motion=use_mm and (2 ** (3 - i) in mm_res),  # reference hard-codes 3 (unet_3d.py:183)
                      resnets=resnets, resample=i != nblk - 1)
        )
    return specs


DEFAULT_UNET3D_CONFIG = dict(  # ctor defaults of the reference (unet_3d.py:34-81)
    sample_size=None, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
    down_block_types=("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
    mid_block_type="UNetMidBlock3DCrossAttn",
    up_block_types=("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"),
    only_cross_attention=False, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    downsample_padding=1, mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-5,
    cross_attention_dim=1280, attention_head_dim=8, dual_cross_attention=False, use_linear_projection=False,
    class_embed_type=None, num_class_embeds=None, upcast_attention=False, resnet_time_scale_shift="default",
    use_inflated_groupnorm=False, use_motion_module=False, motion_module_resolutions=(1, 2, 4, 8),
    motion_module_mid_block=False, motion_module_decoder_only=False, motion_module_type=None,
    motion_module_kwargs={}, unet_use_cross_frame_attention=None, unet_use_temporal_attention=None,
)

SD15_INFERENCE_V2 = dict(  # SD-1.5 unet/config.json + configs/inference/inference_v2.yaml:1-22
    cross_attention_dim=768, use_inflated_groupnorm=True, unet_use_cross_frame_attention=False,
    unet_use_temporal_attention=False, use_motion_module=True, motion_module_resolutions=(1, 2, 4, 8),
    motion_module_mid_block=True, motion_module_decoder_only=False, motion_module_type="Vanilla",
    motion_module_kwargs=dict(num_attention_heads=8, num_transformer_block=1,
                              attention_block_types=("Temporal_Self", "Temporal_Self"),
                              temporal_position_encoding=True, temporal_position_encoding_max_len=32,
                              temporal_attention_dim_div=1),
)
