"""ReferenceNet: drop-in for src.models.unet_2d_condition.UNet2DConditionModel as the CamAnimate
pipeline uses it (/root/reference/src/models/unet_2d_condition.py:872-1308 with conv_norm_out /
conv_out removed, :645-653, 1295-1299): an SD-1.5 UNet run ONCE per clip on the reference latent at
t = 0 whose only purpose is to fill the `bank` of every BasicTransformerBlock with its post-norm1
features (mutual_self_attention.py:137-146).

Same parameter grammar as the SD-1.5 checkpoint (minus conv_norm_out / conv_out); same native
executor as the denoising UNet with one frame per batch entry and no motion modules.
"""
from __future__ import annotations

import json
import os
from typing import List

import torch
from torch import nn

from .arch import BlockSpec, build_block_specs
from .unet3d import (AttentionParams, FeedForwardParams, FrozenConfig, InflatedConv3d, _Holder, _TimestepEmbedding,
                     _UNetBlock)

DEFAULT_UNET2D_CONFIG = dict(  # the SD-1.5 `unet/config.json` fields this path reads
    sample_size=None, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    mid_block_type="UNetMidBlock2DCrossAttn",
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    only_cross_attention=False, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    downsample_padding=1, mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-5,
    cross_attention_dim=1280, attention_head_dim=8, dual_cross_attention=False, use_linear_projection=False,
    class_embed_type=None, num_class_embeds=None, upcast_attention=False, resnet_time_scale_shift="default",
)


class BasicTransformerBlock(_Holder):
    """src/models/attention.py:12-295 parameter layout (norm1/attn1, norm2/attn2, norm3/ff)."""

    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = AttentionParams(dim, None, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = AttentionParams(dim, cross_attention_dim, heads, dim_head)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForwardParams(dim)
        self.bank: List[torch.Tensor] = []


class Transformer2DModel(_Holder):  # src/models/transformer_2d.py
    def __init__(self, heads, dim_head, in_channels, cross_attention_dim, norm_num_groups=32):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim)])
        self.proj_out = nn.Conv2d(inner, in_channels, 1)


class _UNet2DBlock(_UNetBlock):
    def __init__(self, spec: BlockSpec, cfg: dict, temb: int):
        super().__init__(spec, cfg, temb)
        if spec.has_attn:
            heads = cfg["attention_head_dim"]
            n_attn = len(spec.resnets) - 1 if spec.kind == "mid" else len(spec.resnets)
            self.attentions = nn.ModuleList(
                [Transformer2DModel(heads, spec.channels // heads, spec.channels, cfg["cross_attention_dim"],
                                    cfg["norm_num_groups"]) for _ in range(n_attn)]
            )
        del self.motion_modules  # the 2-D blocks have no such child (keeps the state-dict identical)


_TO3D = {"CrossAttnDownBlock2D": "CrossAttnDownBlock3D", "DownBlock2D": "DownBlock3D",
         "CrossAttnUpBlock2D": "CrossAttnUpBlock3D", "UpBlock2D": "UpBlock3D"}


class UNet2DConditionModel(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        cfg = dict(DEFAULT_UNET2D_CONFIG)
        cfg.update({k: v for k, v in kwargs.items() if k in cfg})
        self._internal_dict = FrozenConfig(cfg)
        walk = dict(cfg)
        walk["down_block_types"] = [_TO3D.get(t, t) for t in cfg["down_block_types"]]
        walk["up_block_types"] = [_TO3D.get(t, t) for t in cfg["up_block_types"]]
        walk["mid_block_type"] = "UNetMidBlock3DCrossAttn"
        walk.update(use_motion_module=False, motion_module_kwargs={}, motion_module_type=None)
        boc = tuple(cfg["block_out_channels"])
        temb = boc[0] * 4
        self.conv_in = InflatedConv3d(cfg["in_channels"], boc[0], 3, padding=1)
        self.time_proj = nn.Identity()
        self.time_embedding = _TimestepEmbedding(boc[0], temb)
        self.specs = build_block_specs(walk)
        self.down_blocks = nn.ModuleList([_UNet2DBlock(s, walk, temb) for s in self.specs if s.kind == "down"])
        self.up_blocks = nn.ModuleList([_UNet2DBlock(s, walk, temb) for s in self.specs if s.kind == "up"])
        self.mid_block = _UNet2DBlock([s for s in self.specs if s.kind == "mid"][0], walk, temb)
        self.num_upsamplers = sum(1 for s in self.specs if s.kind == "up" and s.resample)
        self.conv_norm_out = None
        self.conv_act = None
        self._walk_cfg = FrozenConfig(walk)
        self._engine = None
        self._reference_mode = None

    @property
    def config(self):
        return self._internal_dict

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            d = self.__dict__.get("_internal_dict")
            if d is not None and name in d:
                return d[name]
            raise

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    @classmethod
    def from_config(cls, config: dict, **overrides):
        cfg = dict(config)
        cfg.update(overrides)
        return cls(**cfg)

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, **kw):
        path = os.fspath(pretrained_model_path)
        if subfolder is not None:
            path = os.path.join(path, subfolder)
        with open(os.path.join(path, "config.json")) as fh:
            model = cls.from_config(json.load(fh))
        st = os.path.join(path, "diffusion_pytorch_model.safetensors")
        pt = os.path.join(path, "diffusion_pytorch_model.bin")
        if os.path.exists(st):
            from safetensors.torch import load_file

            sd = load_file(st, device="cpu")
        elif os.path.exists(pt):
            sd = torch.load(pt, map_location="cpu", weights_only=True)
        else:
            raise FileNotFoundError(f"no weights file found in {path}")
        model.load_state_dict(sd, strict=False)  # the checkpoint's conv_norm_out / conv_out are unused here
        return model

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def engine(self):
        if self._engine is None:
            from .engine import UNet3DEngine

            self._engine = UNet3DEngine(self, kind="reference")
        return self._engine

    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states, return_dict: bool = True, **unused):
        """sample [b,4,h,w].  In write mode (ReferenceAttentionControl) fills the banks; returns the
        last up-block's hidden state like the reference (its conv_out is commented out)."""
        out = self.engine().forward_reference(sample, timestep, encoder_hidden_states)
        return (out,) if not return_dict else FrozenConfig(sample=out)
