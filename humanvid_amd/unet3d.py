"""Parameter containers with the reference's module/parameter names, and the drop-in
`UNet3DConditionModel` whose forward runs on the native gfx950 executor (`engine.py`).

The nn.Module tree exists for checkpoint compatibility only (state-dict grammar of
/root/reference/src/models/unet_3d.py, unet_3d_blocks.py, resnet.py, transformer_3d.py,
attention.py, motion_module.py -- SURVEY.md appendix B); none of these modules implements an eager
PyTorch forward: compute happens exclusively in libhumanvid_hip.so, and calling a container raises.
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple, Union

import torch
from torch import nn

from .arch import DEFAULT_UNET3D_CONFIG, BlockSpec, build_block_specs


class _Holder(nn.Module):
    """A module that only holds parameters; the native executor consumes them."""

    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError(
            f"{type(self).__name__} is a parameter container: humanvid_amd has no eager PyTorch path; "
            "run the owning model's forward (native HIP executor)."
        )


class FrozenConfig(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


@dataclass
class UNet3DConditionOutput:
    sample: torch.Tensor

    def __getitem__(self, i):
        return (self.sample,)[i]


# ------------------------------------------------------------------------------------ leaf containers
class InflatedConv3d(nn.Conv2d):  # src/models/resnet.py:9-15 (2-D conv applied per frame)
    forward = _Holder.forward


class InflatedGroupNorm(nn.GroupNorm):  # src/models/resnet.py:18-26 (per-frame statistics)
    forward = _Holder.forward


class AttentionParams(_Holder):
    """diffusers `Attention` parameter layout: to_q/to_k/to_v (no bias), to_out = [Linear, Dropout]."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, bias=False):
        super().__init__()
        inner = heads * dim_head
        cross = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.heads, self.dim_head = heads, dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(cross, inner, bias=bias)
        self.to_v = nn.Linear(cross, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])


class _GEGLU(_Holder):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForwardParams(_Holder):
    """diffusers `FeedForward(activation_fn="geglu")`: net = [GEGLU, Dropout, Linear]."""

    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([_GEGLU(dim, 4 * dim), nn.Dropout(0.0), nn.Linear(4 * dim, dim)])


class ResnetBlock3D(_Holder):  # src/models/resnet.py:121-213
    def __init__(self, in_channels, out_channels, temb_channels, groups=32, eps=1e-5):
        super().__init__()
        self.in_channels, self.out_channels, self.eps = in_channels, out_channels, eps
        self.norm1 = InflatedGroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = InflatedConv3d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = InflatedGroupNorm(groups, out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = InflatedConv3d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = InflatedConv3d(in_channels, out_channels, 1) if in_channels != out_channels else None


class Downsample3D(_Holder):  # src/models/resnet.py:91-118
    def __init__(self, channels):
        super().__init__()
        self.conv = InflatedConv3d(channels, channels, 3, stride=2, padding=1)


class Upsample3D(_Holder):  # src/models/resnet.py:29-88
    def __init__(self, channels):
        super().__init__()
        self.conv = InflatedConv3d(channels, channels, 3, padding=1)


class TemporalBasicTransformerBlock(_Holder):
    """src/models/attention.py:298-443.  `bank` is filled by ReferenceAttentionControl.update()."""

    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.attn1 = AttentionParams(dim, None, heads, dim_head)
        self.norm1 = nn.LayerNorm(dim)
        self.attn2 = AttentionParams(dim, cross_attention_dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.ff = FeedForwardParams(dim)
        self.norm3 = nn.LayerNorm(dim)
        self.bank: List[torch.Tensor] = []


class Transformer3DModel(_Holder):  # src/models/transformer_3d.py:27-101
    def __init__(self, heads, dim_head, in_channels, cross_attention_dim, norm_num_groups=32):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList(
            [TemporalBasicTransformerBlock(inner, heads, dim_head, cross_attention_dim)]
        )
        self.proj_out = nn.Conv2d(inner, in_channels, 1)


def sinusoidal_pe(max_len: int, d_model: int) -> torch.Tensor:
    """src/models/motion_module.py:262-273."""
    position = torch.arange(max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(1, max_len, d_model)
    pe[0, :, 0::2] = torch.sin(position * div_term)
    pe[0, :, 1::2] = torch.cos(position * div_term)
    return pe


class PositionalEncoding(_Holder):
    def __init__(self, d_model, max_len=24):
        super().__init__()
        self.register_buffer("pe", sinusoidal_pe(max_len, d_model))


class VersatileAttention(AttentionParams):  # src/models/motion_module.py:280-349
    def __init__(self, dim, heads, dim_head, temporal_position_encoding, max_len):
        super().__init__(dim, None, heads, dim_head)
        self.pos_encoder = PositionalEncoding(dim, max_len) if temporal_position_encoding else None


class TemporalTransformerBlock(_Holder):  # src/models/motion_module.py:185-234
    def __init__(self, dim, heads, dim_head, attention_block_types, temporal_position_encoding, max_len):
        super().__init__()
        for name in attention_block_types:
            if name != "Temporal_Self":
                raise NotImplementedError(f"attention block type {name}")
        self.attention_blocks = nn.ModuleList(
            [VersatileAttention(dim, heads, dim_head, temporal_position_encoding, max_len) for _ in attention_block_types]
        )
        self.norms = nn.ModuleList([nn.LayerNorm(dim) for _ in attention_block_types])
        self.ff = FeedForwardParams(dim)
        self.ff_norm = nn.LayerNorm(dim)


class TemporalTransformer3DModel(_Holder):  # src/models/motion_module.py:94-144
    def __init__(self, in_channels, heads, dim_head, num_layers, attention_block_types, temporal_position_encoding,
                 max_len, norm_num_groups=32):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [TemporalTransformerBlock(inner, heads, dim_head, attention_block_types, temporal_position_encoding, max_len)
             for _ in range(num_layers)]
        )
        self.proj_out = nn.Linear(inner, in_channels)


def zero_module(module):  # src/models/motion_module.py:15-19
    for p in module.parameters():
        p.detach().zero_()
    return module


class VanillaTemporalModule(_Holder):  # src/models/motion_module.py:44-75
    def __init__(self, in_channels, num_attention_heads=8, num_transformer_block=2,
                 attention_block_types=("Temporal_Self", "Temporal_Self"), cross_frame_attention_mode=None,
                 temporal_position_encoding=False, temporal_position_encoding_max_len=24,
                 temporal_attention_dim_div=1, zero_initialize=True):
        super().__init__()
        self.temporal_transformer = TemporalTransformer3DModel(
            in_channels, num_attention_heads, in_channels // num_attention_heads // temporal_attention_dim_div,
            num_transformer_block, tuple(attention_block_types), temporal_position_encoding,
            temporal_position_encoding_max_len,
        )
        if zero_initialize:
            self.temporal_transformer.proj_out = zero_module(self.temporal_transformer.proj_out)


def get_motion_module(in_channels, motion_module_type: str, motion_module_kwargs: dict):
    if motion_module_type == "Vanilla":
        return VanillaTemporalModule(in_channels=in_channels, **motion_module_kwargs)
    raise ValueError


class _UNetBlock(_Holder):
    """One down / mid / up block: resnets (+ attentions) (+ motion_modules) (+ down/upsamplers)."""

    def __init__(self, spec: BlockSpec, cfg: dict, temb: int):
        super().__init__()
        heads = cfg["attention_head_dim"]
        heads = heads if isinstance(heads, int) else heads[0]
        groups, eps = cfg["norm_num_groups"], cfg["norm_eps"]
        self.has_cross_attention = spec.has_attn
        n_attn = len(spec.resnets) - 1 if spec.kind == "mid" else len(spec.resnets)
        self.resnets = nn.ModuleList([ResnetBlock3D(m + s, o, temb, groups, eps) for (m, s, o) in spec.resnets])
        if spec.has_attn:
            self.attentions = nn.ModuleList(
                [Transformer3DModel(heads, spec.channels // heads, spec.channels, cfg["cross_attention_dim"], groups)
                 for _ in range(n_attn)]
            )
        mm = [
            get_motion_module(spec.channels, cfg["motion_module_type"], dict(cfg["motion_module_kwargs"]))
            if spec.has_motion else None
            for _ in range(n_attn)
        ]
        self.motion_modules = nn.ModuleList(mm)
        if spec.kind == "down":
            self.downsamplers = nn.ModuleList([Downsample3D(spec.channels)]) if spec.resample else None
        if spec.kind == "up":
            self.upsamplers = nn.ModuleList([Upsample3D(spec.channels)]) if spec.resample else None


class CrossAttnDownBlock3D(_UNetBlock):
    pass


class DownBlock3D(_UNetBlock):
    pass


class UNetMidBlock3DCrossAttn(_UNetBlock):
    pass


class CrossAttnUpBlock3D(_UNetBlock):
    pass


class UpBlock3D(_UNetBlock):
    pass


_BLOCK_CLASSES = {
    "CrossAttnDownBlock3D": CrossAttnDownBlock3D, "DownBlock3D": DownBlock3D,
    "CrossAttnUpBlock3D": CrossAttnUpBlock3D, "UpBlock3D": UpBlock3D,
}


class _TimestepEmbedding(_Holder):
    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)


class UNet3DConditionModel(nn.Module):
    """Drop-in for src.models.unet_3d.UNet3DConditionModel (ctor kwargs unet_3d.py:34-81,
    forward :397-408, from_pretrained_2d :580-587).  forward() executes on the MI355X through
    libhumanvid_hip.so; there is no PyTorch fallback."""

    def __init__(self, **kwargs):
        super().__init__()
        cfg = dict(DEFAULT_UNET3D_CONFIG)
        unknown = set(kwargs) - set(cfg)
        if unknown:
            raise TypeError(f"unexpected config keys: {sorted(unknown)}")
        cfg.update(kwargs)
        cfg["motion_module_kwargs"] = dict(cfg["motion_module_kwargs"] or {})
        if cfg["use_motion_module"] and not cfg["use_inflated_groupnorm"]:
            raise NotImplementedError(
                "use_inflated_groupnorm=False couples GroupNorm statistics across frames (inference_v1.yaml); "
                "only the inference_v2 geometry is supported by the native path"
            )
        if cfg["class_embed_type"] is not None or cfg["num_class_embeds"] is not None:
            raise NotImplementedError("class embeddings are not part of the CamAnimate path")
        self._internal_dict = FrozenConfig(cfg)
        boc = tuple(cfg["block_out_channels"])
        temb = boc[0] * 4
        self.sample_size = cfg["sample_size"]
        self.conv_in = InflatedConv3d(cfg["in_channels"], boc[0], 3, padding=(1, 1))
        self.time_proj = nn.Identity()  # parameter-free sinusoid (computed by hv_timestep_embedding)
        self.time_embedding = _TimestepEmbedding(boc[0], temb)
        self.specs = build_block_specs(cfg)
        self.down_blocks = nn.ModuleList()
        self.up_blocks = nn.ModuleList()
        self.mid_block = None
        types = list(cfg["down_block_types"]) + ["mid"] + list(cfg["up_block_types"])
        for spec, btype in zip(self.specs, types):
            if spec.kind == "down":
                self.down_blocks.append(_BLOCK_CLASSES[btype](spec, cfg, temb))
        # registration order down, up, mid matters: ReferenceAttentionControl pairs banks in DFS order
        for spec, btype in zip(self.specs, types):
            if spec.kind == "up":
                self.up_blocks.append(_BLOCK_CLASSES[btype](spec, cfg, temb))
        mid_spec = [s for s in self.specs if s.kind == "mid"][0]
        self.mid_block = UNetMidBlock3DCrossAttn(mid_spec, cfg, temb)
        self.num_upsamplers = sum(1 for s in self.specs if s.kind == "up" and s.resample)
        self.conv_norm_out = InflatedGroupNorm(cfg["norm_num_groups"], boc[0], eps=cfg["norm_eps"])
        self.conv_act = nn.SiLU()
        self.conv_out = InflatedConv3d(boc[0], cfg["out_channels"], 3, padding=1)
        self._engine = None
        self._reference_mode: Optional[dict] = None  # set by ReferenceAttentionControl(mode="read")

    # ---- diffusers ModelMixin / ConfigMixin surface the callers rely on -------------------------
    @property
    def config(self) -> FrozenConfig:
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
        known = set(DEFAULT_UNET3D_CONFIG)
        cfg = {k: v for k, v in dict(config).items() if k in known}
        cfg.update({k: v for k, v in overrides.items() if k in known})
        return cls(**cfg)

    @classmethod
    def from_pretrained_2d(cls, pretrained_model_path, motion_module_path, subfolder=None,
                           unet_additional_kwargs=None, mm_zero_proj_out=False):
        """unet_3d.py:579-670: SD-1.5 `unet/config.json` + 2-D weights + motion-module checkpoint,
        merged with strict=False."""
        path = os.fspath(pretrained_model_path)
        if subfolder is not None:
            path = os.path.join(path, subfolder)
        config_file = os.path.join(path, "config.json")
        if not os.path.isfile(config_file):
            raise RuntimeError(f"{config_file} does not exist or is not a file")
        with open(config_file) as fh:
            unet_config = json.load(fh)
        unet_config["down_block_types"] = ["CrossAttnDownBlock3D"] * 3 + ["DownBlock3D"]
        unet_config["up_block_types"] = ["UpBlock3D"] + ["CrossAttnUpBlock3D"] * 3
        unet_config["mid_block_type"] = "UNetMidBlock3DCrossAttn"
        model = cls.from_config(unet_config, **dict(unet_additional_kwargs or {}))
        st = os.path.join(path, "diffusion_pytorch_model.safetensors")
        pt = os.path.join(path, "diffusion_pytorch_model.bin")
        if os.path.exists(st):
            from safetensors.torch import load_file

            state_dict = load_file(st, device="cpu")
        elif os.path.exists(pt):
            state_dict = torch.load(pt, map_location="cpu", weights_only=True)
        else:
            raise FileNotFoundError(f"no weights file found in {path}")
        mm_path = os.fspath(motion_module_path)
        if os.path.isfile(mm_path):
            ext = os.path.splitext(mm_path)[1].lower()
            if ext in (".pth", ".pt", ".ckpt"):
                motion_sd = torch.load(mm_path, map_location="cpu", weights_only=True)
            elif ext == ".safetensors":
                from safetensors.torch import load_file

                motion_sd = load_file(mm_path, device="cpu")
            else:
                raise RuntimeError(f"unknown file format for motion module weights: {ext}")
            if mm_zero_proj_out:
                motion_sd = {k: v for k, v in motion_sd.items() if "proj_out" not in k}
            state_dict.update(motion_sd)
        model.load_state_dict(state_dict, strict=False)
        return model

    def _apply(self, fn, *a, **k):  # .to()/.half()/.cuda() invalidate the packed device weights
        self._engine = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    # ---- native execution ------------------------------------------------------------------------
    def engine(self):
        if self._engine is None:
            from .engine import UNet3DEngine

            self._engine = UNet3DEngine(self)
        return self._engine

    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, pose_cond_fea=None,
                attention_mask=None, down_block_additional_residuals=None, mid_block_additional_residual=None,
                return_dict: bool = True) -> Union[UNet3DConditionOutput, Tuple]:
        if attention_mask is not None or down_block_additional_residuals is not None or \
                mid_block_additional_residual is not None or class_labels is not None:
            raise NotImplementedError("attention_mask / additional residuals / class_labels are unused on this path")
        out = self.engine().forward_ncfhw(sample, timestep, encoder_hidden_states, pose_cond_fea)
        if not return_dict:
            return (out,)
        return UNet3DConditionOutput(sample=out)


def transformer_locations(unet: UNet3DConditionModel) -> List[str]:
    """Module names of the spatial transformer blocks in the order ReferenceAttentionControl pairs
    reader and writer: DFS order, stably sorted by descending hidden size
    (src/models/mutual_self_attention.py:267-287)."""
    blocks = [(n, m) for n, m in unet.named_modules() if isinstance(m, TemporalBasicTransformerBlock)]
    blocks = sorted(blocks, key=lambda nm: -nm[1].norm1.normalized_shape[0])
    return [n.rsplit(".transformer_blocks.0", 1)[0] for n, _ in blocks]
