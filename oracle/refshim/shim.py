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
    def __init__(self, in_channels, time_embed_dim, act_fn="silu"):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample):
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
        SinusoidalPositionalEmbedding=SinusoidalPositionalEmbedding)
    mod("diffusers.models.attention", AdaLayerNorm=AdaLayerNorm, Attention=Attention, FeedForward=FeedForward, GEGLU=GEGLU)
    mod("diffusers.models.attention_processor", Attention=Attention, AttnProcessor=AttnProcessor,
        AttnProcessor2_0=AttnProcessor2_0, AttentionProcessor=AttentionProcessor)
    mod("diffusers.models.lora", LoRALinearLayer=LoRALinearLayer)
    # imported (never instantiated) by src/cameractrl/resnet.py, which the camera encoder pulls in
    mod("diffusers.models.activations", get_activation=lambda name: nn.SiLU())
    mod("diffusers.models.normalization", AdaGroupNorm=AdaLayerNorm)
    sys.modules["diffusers.models.attention_processor"].SpatialNorm = AdaLayerNorm
    u = mod("diffusers.utils", BaseOutput=BaseOutput, logging=_Logging, SAFETENSORS_WEIGHTS_NAME=SAFETENSORS_WEIGHTS_NAME,
            WEIGHTS_NAME=WEIGHTS_NAME, USE_PEFT_BACKEND=USE_PEFT_BACKEND, is_torch_version=is_torch_version)
    u.__path__ = []
    mod("diffusers.utils.import_utils", is_xformers_available=is_xformers_available)
