"""Minimal restatement of the diffusers==0.24.0 symbols the reference's hot path imports.

TEST INFRASTRUCTURE ONLY.  `diffusers` (pinned 0.24.0 in /root/reference/environment.yml:87)
is not installed in this image and cannot be installed (no network).  To run the reference's
*own* model code (imported verbatim from /root/reference) as the ground truth, `install()`
registers stand-ins for exactly the symbols those files import.  Semantics follow the
published diffusers 0.24.0 release; they could not be diffed against
the real package here, so parity towards diffusers itself is "unpinned" and this file is the
specification.  Nothing here is ever loaded by the product package.
"""
from __future__ import annotations

import functools
import inspect
import logging as _pylogging
import math
import sys
import types
from collections import OrderedDict
from dataclasses import fields, is_dataclass

import torch
import torch.nn.functional as F
from torch import nn


# ---- diffusers.utils -----------------------------------------------------------------
class BaseOutput(OrderedDict):
    """dataclass-backed ordered dict (diffusers.utils.BaseOutput)."""

    def __post_init__(self):
        for f in fields(self):
            v = getattr(self, f.name)
            if v is not None:
                self[f.name] = v

    def __getitem__(self, k):
        if isinstance(k, str):
            return dict(self.items())[k]
        return self.to_tuple()[k]

    def to_tuple(self):
        return tuple(self[k] for k in self.keys())


class _Logging:
    @staticmethod
    def get_logger(name):
        return _pylogging.getLogger(name)


SAFETENSORS_WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"
WEIGHTS_NAME = "diffusion_pytorch_model.bin"
USE_PEFT_BACKEND = False


def is_xformers_available():
    return False


def is_torch_version(op, ver):
    return True


# ---- configuration_utils / modeling_utils ------------------------------------------------
class FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        sig = inspect.signature(init)
        params = list(sig.parameters.items())[1:]
        cfg = {n: p.default for n, p in params if p.default is not inspect.Parameter.empty}
        for (n, _), a in zip(params, args):
            cfg[n] = a
        cfg.update(kwargs)
        init(self, *args, **kwargs)
        self._internal_dict = FrozenDict(cfg)

    return inner


class ConfigMixin:
    config_name = "config.json"

    @property
    def config(self):
        return self._internal_dict

    def __getattr__(self, name):
        # diffusers falls back to the config dict for unknown attributes
        if "_internal_dict" in self.__dict__ and name in self.__dict__["_internal_dict"]:
            return self.__dict__["_internal_dict"][name]
        return super().__getattr__(name)

    @classmethod
    def from_config(cls, config, **kwargs):
        sig = inspect.signature(cls.__init__)
        cfg = {k: v for k, v in dict(config).items() if k in sig.parameters}
        cfg.update({k: v for k, v in kwargs.items() if k in sig.parameters})
        return cls(**cfg)


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


# ---- embeddings ------------------------------------------------------------------------
class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels, self.flip, self.shift = num_channels, flip_sin_to_cos, downscale_freq_shift

    def forward(self, timesteps):
        half = self.num_channels // 2
        exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32, device=timesteps.device)
        exponent = exponent / (half - self.shift)
        emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
        if self.flip:
            emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
        return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None, post_act_fn=None, cond_proj_dim=None):
        super().__init__()
        if out_dim is not None or post_act_fn is not None or cond_proj_dim is not None:
            raise NotImplementedError("only the SD-1.5 form of TimestepEmbedding is restated")
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


class SinusoidalPositionalEmbedding(nn.Module):  # imported, never instantiated on this path
    pass


class AdaLayerNorm(nn.Module):  # imported, never instantiated on this path
    pass


# ---- attention -------------------------------------------------------------------------
class AttnProcessor2_0:
    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        assert attention_mask is None
        q = attn.to_q(hidden_states)
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        k, v = attn.to_k(ctx), attn.to_v(ctx)
        B, L, inner = q.shape
        d = inner // attn.heads
        q = q.view(B, -1, attn.heads, d).transpose(1, 2)
        k = k.view(B, -1, attn.heads, d).transpose(1, 2)
        v = v.view(B, -1, attn.heads, d).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(B, -1, inner).to(q.dtype)
        o = attn.to_out[0](o)
        o = attn.to_out[1](o)
        return o / attn.rescale_output_factor


class AttnProcessor(AttnProcessor2_0):
    """classic baddbmm+softmax processor: same mathematics as SDPA."""


AttentionProcessor = AttnProcessor2_0


class Attention(nn.Module):
    def __init__(
        self,
        query_dim,
        cross_attention_dim=None,
        heads=8,
        dim_head=64,
        dropout=0.0,
        bias=False,
        upcast_attention=False,
        upcast_softmax=False,
        rescale_output_factor=1.0,
        **unused,
    ):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.scale = dim_head**-0.5
        self.rescale_output_factor = rescale_output_factor
        cross = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(cross, inner, bias=bias)
        self.to_v = nn.Linear(cross, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])
        self.processor = AttnProcessor2_0()

    def set_processor(self, p):
        self.processor = p

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(
            self, hidden_states, encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask, **kw
        )


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, g = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(g)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False):
        super().__init__()
        assert activation_fn == "geglu"
        inner = dim * mult
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim_out or dim)])

    def forward(self, x, scale=1.0):
        for m in self.net:
            x = m(x)
        return x


class LoRALinearLayer(nn.Module):  # imported by cameractrl/attention_processor.py, unused
    def __init__(self, *a, **k):
        super().__init__()


# ---- diffusers 2-D UNet building blocks (diffusers==0.24.0: models/resnet.py, models/lora.py) -----------------------
# Needed so that the reference's ReferenceNet (/root/reference/src/models/unet_2d_condition.py, unet_2d_blocks.py,
# transformer_2d.py) imports and runs in write mode.  Only the code paths SD-1.5 takes are implemented; everything else
# raises.  Semantics restated from the published 0.24.0 sources.
class LoRACompatibleConv(nn.Conv2d):
    def forward(self, hidden_states, scale: float = 1.0):
        return super().forward(hidden_states)


class LoRACompatibleLinear(nn.Linear):
    def forward(self, hidden_states, scale: float = 1.0):
        return super().forward(hidden_states)


def get_activation(name: str):
    name = name.lower()
    if name in ("swish", "silu"):
        return nn.SiLU()
    if name == "mish":
        return nn.Mish()
    if name == "gelu":
        return nn.GELU()
    if name == "relu":
        return nn.ReLU()
    raise ValueError(f"Unsupported activation function: {name}")


class Upsample2D(nn.Module):
    """nearest x2 (+ 3x3 conv): models/resnet.py Upsample2D with use_conv=True, use_conv_transpose=False"""

    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None, name="conv"):
        super().__init__()
        if use_conv_transpose or not use_conv:
            raise NotImplementedError("only the SD-1.5 form (nearest + conv) is restated")
        self.channels, self.out_channels, self.name = channels, out_channels or channels, name
        conv = LoRACompatibleConv(self.channels, self.out_channels, 3, padding=1)
        if name == "conv":
            self.conv = conv
        else:
            self.Conv2d_0 = conv

    def forward(self, hidden_states, output_size=None, scale: float = 1.0):
        assert hidden_states.shape[1] == self.channels
        dtype = hidden_states.dtype
        if dtype == torch.bfloat16:
            hidden_states = hidden_states.to(torch.float32)
        if hidden_states.shape[0] >= 64:
            hidden_states = hidden_states.contiguous()
        if output_size is None:
            hidden_states = F.interpolate(hidden_states, scale_factor=2.0, mode="nearest")
        else:
            hidden_states = F.interpolate(hidden_states, size=output_size, mode="nearest")
        if dtype == torch.bfloat16:
            hidden_states = hidden_states.to(dtype)
        conv = self.conv if self.name == "conv" else self.Conv2d_0
        return conv(hidden_states, scale)


class Downsample2D(nn.Module):
    """3x3 stride-2 conv: models/resnet.py Downsample2D with use_conv=True"""

    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv"):
        super().__init__()
        if not use_conv:
            raise NotImplementedError("only the SD-1.5 form (strided conv) is restated")
        self.channels, self.out_channels, self.padding, self.name = channels, out_channels or channels, padding, name
        conv = LoRACompatibleConv(self.channels, self.out_channels, 3, stride=2, padding=padding)
        if name == "conv":
            self.Conv2d_0 = conv
            self.conv = conv
        else:  # "Conv2d_0" and every other name (the UNet blocks pass "op") register the layer as `conv`
            self.conv = conv

    def forward(self, hidden_states, scale: float = 1.0):
        assert hidden_states.shape[1] == self.channels
        if self.padding == 0:
            hidden_states = F.pad(hidden_states, (0, 1, 0, 1), mode="constant", value=0)
        return self.conv(hidden_states, scale)


class ResnetBlock2D(nn.Module):
    """models/resnet.py ResnetBlock2D, time_embedding_norm="default", no up/down, kernel None"""

    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512, groups=32,
                 groups_out=None, pre_norm=True, eps=1e-6, non_linearity="swish", skip_time_act=False,
                 time_embedding_norm="default", kernel=None, output_scale_factor=1.0, use_in_shortcut=None, up=False,
                 down=False, conv_shortcut_bias=True, conv_2d_out_channels=None):
        super().__init__()
        if time_embedding_norm != "default" or up or down or kernel is not None:
            raise NotImplementedError("only the SD-1.5 form of ResnetBlock2D is restated")
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels, self.output_scale_factor = in_channels, out_channels, output_scale_factor
        self.skip_time_act = skip_time_act
        groups_out = groups if groups_out is None else groups_out
        self.norm1 = nn.GroupNorm(num_groups=groups, num_channels=in_channels, eps=eps, affine=True)
        self.conv1 = LoRACompatibleConv(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.time_emb_proj = LoRACompatibleLinear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(num_groups=groups_out, num_channels=out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(dropout)
        conv_2d_out_channels = conv_2d_out_channels or out_channels
        self.conv2 = LoRACompatibleConv(out_channels, conv_2d_out_channels, kernel_size=3, stride=1, padding=1)
        self.nonlinearity = get_activation(non_linearity)
        self.use_in_shortcut = in_channels != conv_2d_out_channels if use_in_shortcut is None else use_in_shortcut
        self.conv_shortcut = None
        if self.use_in_shortcut:
            self.conv_shortcut = LoRACompatibleConv(in_channels, conv_2d_out_channels, kernel_size=1, stride=1, padding=0,
                                                    bias=conv_shortcut_bias)

    def forward(self, input_tensor, temb, scale: float = 1.0):
        hidden_states = self.nonlinearity(self.norm1(input_tensor))
        hidden_states = self.conv1(hidden_states, scale)
        if self.time_emb_proj is not None:
            if not self.skip_time_act:
                temb = self.nonlinearity(temb)
            temb = self.time_emb_proj(temb, scale)[:, :, None, None]
        if temb is not None:
            hidden_states = hidden_states + temb
        hidden_states = self.nonlinearity(self.norm2(hidden_states))
        hidden_states = self.conv2(self.dropout(hidden_states), scale)
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor, scale)
        return (input_tensor + hidden_states) / self.output_scale_factor


class _Unavailable(nn.Module):
    """placeholder for symbols the reference imports but SD-1.5 never instantiates"""

    def __init__(self, *a, **k):
        raise NotImplementedError(f"{type(self).__name__} is not part of the SD-1.5 ReferenceNet path")


def _unavailable(name):
    return type(name, (_Unavailable,), {})


class UNet2DConditionLoadersMixin:
    pass


def deprecate(*args, **kwargs):
    return None


def scale_lora_layers(model, weight):
    return None


def unscale_lora_layers(model, weight=None):
    return None


def apply_freeu(resolution_idx, hidden_states, res_hidden_states, **freeu_kwargs):
    return hidden_states, res_hidden_states


def install():
    """Register the stand-ins under the module paths the reference imports."""

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    d = mod("diffusers")
    d.__path__ = []  # mark as package
    mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config, FrozenDict=FrozenDict)
    mm = mod("diffusers.models", ModelMixin=ModelMixin)
    mm.__path__ = []
    mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    mod("diffusers.models.embeddings", TimestepEmbedding=TimestepEmbedding, Timesteps=Timesteps,
        SinusoidalPositionalEmbedding=SinusoidalPositionalEmbedding,
        **{n: _unavailable(n) for n in ("GaussianFourierProjection", "ImageHintTimeEmbedding", "ImageProjection",
                                        "ImageTimeEmbedding", "PositionNet", "TextImageProjection", "TextImageTimeEmbedding",
                                        "TextTimeEmbedding", "CaptionProjection")})
    mod("diffusers.models.attention", AdaLayerNorm=AdaLayerNorm, Attention=Attention, FeedForward=FeedForward, GEGLU=GEGLU)
    mod("diffusers.models.attention_processor", Attention=Attention, AttnProcessor=AttnProcessor,
        AttnProcessor2_0=AttnProcessor2_0, AttentionProcessor=AttentionProcessor,
        AttnAddedKVProcessor=_unavailable("AttnAddedKVProcessor"), ADDED_KV_ATTENTION_PROCESSORS=(),
        CROSS_ATTENTION_PROCESSORS=(AttnProcessor, AttnProcessor2_0))
    mod("diffusers.models.lora", LoRALinearLayer=LoRALinearLayer, LoRACompatibleConv=LoRACompatibleConv,
        LoRACompatibleLinear=LoRACompatibleLinear)
    mod("diffusers.models.activations", get_activation=get_activation)
    mod("diffusers.models.normalization", AdaGroupNorm=AdaLayerNorm, AdaLayerNormSingle=_unavailable("AdaLayerNormSingle"))
    mod("diffusers.models.resnet", ResnetBlock2D=ResnetBlock2D, Downsample2D=Downsample2D, Upsample2D=Upsample2D)
    mod("diffusers.models.dual_transformer_2d", DualTransformer2DModel=_unavailable("DualTransformer2DModel"))
    mod("diffusers.loaders", UNet2DConditionLoadersMixin=UNet2DConditionLoadersMixin)
    sys.modules["diffusers.models.attention_processor"].SpatialNorm = AdaLayerNorm
    u = mod("diffusers.utils", BaseOutput=BaseOutput, logging=_Logging, SAFETENSORS_WEIGHTS_NAME=SAFETENSORS_WEIGHTS_NAME,
            WEIGHTS_NAME=WEIGHTS_NAME, USE_PEFT_BACKEND=USE_PEFT_BACKEND, is_torch_version=is_torch_version,
            deprecate=deprecate, scale_lora_layers=scale_lora_layers, unscale_lora_layers=unscale_lora_layers)
    u.__path__ = []
    mod("diffusers.utils.torch_utils", apply_freeu=apply_freeu)
    mod("diffusers.utils.import_utils", is_xformers_available=is_xformers_available)
