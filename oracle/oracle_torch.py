"""CPU fp32 oracle for the CamAnimate denoising path  --  TEST INFRASTRUCTURE ONLY.

This file is a *restatement* (plain PyTorch, CPU, float32, functional style) of the
algorithm the reference executes on the hot path.  It exists so that the HIP path can
be checked on the GPU box, where /root/reference does not exist.  It must never be
imported by the product package (`humanvid_amd`): only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may use it.

Pinning status: the reference ships no tests / golden vectors ("parity unpinned" w.r.t.
upstream).  What pins this file is `oracle/gen_golden.py`: it imports the reference's own
model code from /root/reference (on top of `oracle/refshim`, a restatement of the
un-vendored diffusers==0.24.0 symbols) and checks every function below against it; the
resulting vectors are committed under tests/golden/.

Weights are addressed with the reference's state-dict key names, so
the same dict can be loaded into the reference modules (`load_state_dict(strict=True)`).

All file:line citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

# --------------------------------------------------------------------------------------
# configuration helpers
# --------------------------------------------------------------------------------------

SD15_UNET3D_CFG = dict(  # SD-1.5 geometry + configs/inference/inference_v2.yaml:1-22
    in_channels=4,
    out_channels=4,
    down_block_types=("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
    up_block_types=("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"),
    block_out_channels=(320, 640, 1280, 1280),
    layers_per_block=2,
    norm_num_groups=32,
    norm_eps=1e-5,
    cross_attention_dim=768,
    attention_head_dim=8,  # = number of heads (src/models/unet_3d_blocks.py:353-356)
    use_motion_module=True,
    motion_module_resolutions=(1, 2, 4, 8),
    motion_module_mid_block=True,
    motion_module_kwargs=dict(
        num_attention_heads=8,
        num_transformer_block=1,
        attention_block_types=("Temporal_Self", "Temporal_Self"),
        temporal_position_encoding=True,
        temporal_position_encoding_max_len=32,
    ),
)

CAMERA_ENCODER_CFG = dict(  # configs/inference/inference_v2.yaml:38-50
    downscale_factor=8,
    channels=(320,),
    nums_rb=2,
    cin=384,
    ksize=1,
    sk=True,
    temporal_attention_nhead=8,
    attention_block_types=("Temporal_Self",),
    temporal_position_encoding=True,
    temporal_position_encoding_max_len=24,
)

POSE_GUIDER_CFG = dict(  # scripts/pose2vid.py:137-140
    conditioning_embedding_channels=320,
    conditioning_channels=3,
    block_out_channels=(16, 32, 96, 256),
)


def tiny_unet3d_cfg(**over) -> dict:
    """A 2-level UNet with production channel widths (head dims 40/80) for fast parity runs."""
    cfg = dict(SD15_UNET3D_CFG)
    cfg.update(
        down_block_types=("CrossAttnDownBlock3D", "DownBlock3D"),
        up_block_types=("UpBlock3D", "CrossAttnUpBlock3D"),
        block_out_channels=(320, 640),
        layers_per_block=1,
    )
    cfg.update(over)
    return cfg


# --------------------------------------------------------------------------------------
# architecture walk (restates the constructors: src/models/unet_3d.py:83-248,
# src/models/unet_3d_blocks.py:296-396,467-538,171-267,586-680,749-814)
# --------------------------------------------------------------------------------------


def unet3d_spec(cfg: dict) -> dict:
    boc = tuple(cfg["block_out_channels"])
    nblk = len(boc)
    lpb = cfg["layers_per_block"]
    use_mm = cfg.get("use_motion_module", False)
    mm_res = tuple(cfg.get("motion_module_resolutions", (1, 2, 4, 8)))
    down = []
    out_ch = boc[0]
    for i, btype in enumerate(cfg["down_block_types"]):
        in_ch, out_ch = out_ch, boc[i]
        final = i == nblk - 1
        resnets = [(in_ch if j == 0 else out_ch, out_ch) for j in range(lpb)]
        down.append(
            dict(
                prefix=f"down_blocks.{i}",
                attn=btype == "CrossAttnDownBlock3D",
                motion=use_mm and (2**i in mm_res) and not cfg.get("motion_module_decoder_only", False),
                resnets=resnets,
                channels=out_ch,
                downsample=not final,
            )
        )
    mid = dict(
        prefix="mid_block",
        channels=boc[-1],
        motion=use_mm and cfg.get("motion_module_mid_block", False),
    )
    up = []
    rev = list(reversed(boc))
    out_ch = rev[0]
    for i, btype in enumerate(cfg["up_block_types"]):
        prev_out = out_ch
        out_ch = rev[i]
        in_ch = rev[min(i + 1, nblk - 1)]
        final = i == nblk - 1
        nl = lpb + 1
        resnets = []
        for j in range(nl):
            skip = in_ch if j == nl - 1 else out_ch
            rin = prev_out if j == 0 else out_ch
            resnets.append((rin, skip, out_ch))
        # NB: the reference hard-codes res = 2 ** (3 - i) (src/models/unet_3d.py:183)
        up.append(
            dict(
                prefix=f"up_blocks.{i}",
                attn=btype == "CrossAttnUpBlock3D",
                motion=use_mm and (2 ** (3 - i) in mm_res),
                resnets=resnets,
                channels=out_ch,
                upsample=not final,
            )
        )
    return dict(down=down, mid=mid, up=up, boc=boc)


def transformer_locations(cfg: dict) -> List[str]:
    """Prefixes of every spatial transformer block, in the order ReferenceAttentionControl
    pairs reader and writer banks: module DFS order (down, up, mid -- src/models/unet_3d.py:
    108-110,157) stably sorted by descending hidden size (mutual_self_attention.py:285-287)."""
    spec = unet3d_spec(cfg)
    locs = []
    for blk in spec["down"]:
        if blk["attn"]:
            for j in range(len(blk["resnets"])):
                locs.append((f"{blk['prefix']}.attentions.{j}", blk["channels"]))
    for blk in spec["up"]:
        if blk["attn"]:
            for j in range(len(blk["resnets"])):
                locs.append((f"{blk['prefix']}.attentions.{j}", blk["channels"]))
    locs.append(("mid_block.attentions.0", spec["mid"]["channels"]))
    locs = sorted(locs, key=lambda t: -t[1])  # python sort is stable, like the reference's
    return [p for p, _ in locs]


# --------------------------------------------------------------------------------------
# synthetic weights with the reference's state-dict grammar
# --------------------------------------------------------------------------------------


class _Init:
    def __init__(self, seed: int):
        self.g = torch.Generator().manual_seed(seed)
        self.sd: SD = {}

    def w(self, name, *shape, fan_in=None, gain=1.0):
        fan_in = fan_in or int(np.prod(shape[1:]))
        self.sd[name] = torch.randn(*shape, generator=self.g) * (gain / math.sqrt(fan_in))

    def b(self, name, n, scale=0.05):
        self.sd[name] = torch.randn(n, generator=self.g) * scale

    def norm(self, p, n):
        self.sd[p + ".weight"] = 1.0 + 0.1 * torch.randn(n, generator=self.g)
        self.sd[p + ".bias"] = 0.05 * torch.randn(n, generator=self.g)

    def conv(self, p, cout, cin, k, bias=True):
        self.w(p + ".weight", cout, cin, k, k)
        if bias:
            self.b(p + ".bias", cout)

    def lin(self, p, cout, cin, bias=True):
        self.w(p + ".weight", cout, cin)
        if bias:
            self.b(p + ".bias", cout)

    def attn(self, p, dim, ctx_dim=None):
        ctx_dim = ctx_dim or dim
        self.lin(p + ".to_q", dim, dim, bias=False)
        self.lin(p + ".to_k", dim, ctx_dim, bias=False)
        self.lin(p + ".to_v", dim, ctx_dim, bias=False)
        self.lin(p + ".to_out.0", dim, dim)

    def ff(self, p, dim):
        self.lin(p + ".net.0.proj", 8 * dim, dim)
        self.lin(p + ".net.2", dim, 4 * dim)

    def resnet(self, p, cin, cout, temb=1280):
        self.norm(p + ".norm1", cin)
        self.conv(p + ".conv1", cout, cin, 3)
        self.lin(p + ".time_emb_proj", cout, temb)
        self.norm(p + ".norm2", cout)
        self.conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            self.conv(p + ".conv_shortcut", cout, cin, 1)

    def transformer3d(self, p, c, xdim):
        self.norm(p + ".norm", c)
        self.conv(p + ".proj_in", c, c, 1)
        self.conv(p + ".proj_out", c, c, 1)
        t = p + ".transformer_blocks.0"
        for n in ("norm1", "norm2", "norm3"):
            self.norm(f"{t}.{n}", c)
        self.attn(t + ".attn1", c)
        self.attn(t + ".attn2", c, xdim)
        self.ff(t + ".ff", c)

    def motion(self, p, c, mmk):
        t = p + ".temporal_transformer"
        self.norm(t + ".norm", c)
        self.lin(t + ".proj_in", c, c)
        self.lin(t + ".proj_out", c, c)  # zero-init in the reference; re-randomised here
        for li in range(mmk.get("num_transformer_block", 1)):
            b = f"{t}.transformer_blocks.{li}"
            for ai, _ in enumerate(mmk["attention_block_types"]):
                self.attn(f"{b}.attention_blocks.{ai}", c)
                if mmk.get("temporal_position_encoding", False):
                    self.sd[f"{b}.attention_blocks.{ai}.pos_encoder.pe"] = sinusoidal_pe(
                        mmk["temporal_position_encoding_max_len"], c
                    )
                self.norm(f"{b}.norms.{ai}", c)
            self.ff(b + ".ff", c)
            self.norm(b + ".ff_norm", c)


def sinusoidal_pe(max_len: int, d_model: int) -> Tensor:
    """src/models/motion_module.py:262-273 (and src/cameractrl/motion_module.py:302-316)."""
    position = torch.arange(max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(1, max_len, d_model)
    pe[0, :, 0::2] = torch.sin(position * div_term)
    pe[0, :, 1::2] = torch.cos(position * div_term)
    return pe


def make_unet3d_weights(cfg: dict, seed: int = 0) -> SD:
    ini = _Init(seed)
    spec = unet3d_spec(cfg)
    boc = spec["boc"]
    temb = boc[0] * 4
    xdim = cfg["cross_attention_dim"]
    mmk = cfg.get("motion_module_kwargs", {})
    ini.conv("conv_in", boc[0], cfg["in_channels"], 3)
    ini.lin("time_embedding.linear_1", temb, boc[0])
    ini.lin("time_embedding.linear_2", temb, temb)
    for blk in spec["down"]:
        p = blk["prefix"]
        for j, (ci, co) in enumerate(blk["resnets"]):
            ini.resnet(f"{p}.resnets.{j}", ci, co, temb)
            if blk["attn"]:
                ini.transformer3d(f"{p}.attentions.{j}", co, xdim)
            if blk["motion"]:
                ini.motion(f"{p}.motion_modules.{j}", co, mmk)
        if blk["downsample"]:
            ini.conv(f"{p}.downsamplers.0.conv", blk["channels"], blk["channels"], 3)
    c = spec["mid"]["channels"]
    ini.resnet("mid_block.resnets.0", c, c, temb)
    ini.transformer3d("mid_block.attentions.0", c, xdim)
    if spec["mid"]["motion"]:
        ini.motion("mid_block.motion_modules.0", c, mmk)
    ini.resnet("mid_block.resnets.1", c, c, temb)
    for blk in spec["up"]:
        p = blk["prefix"]
        for j, (rin, skip, co) in enumerate(blk["resnets"]):
            ini.resnet(f"{p}.resnets.{j}", rin + skip, co, temb)
            if blk["attn"]:
                ini.transformer3d(f"{p}.attentions.{j}", co, xdim)
            if blk["motion"]:
                ini.motion(f"{p}.motion_modules.{j}", co, mmk)
        if blk["upsample"]:
            ini.conv(f"{p}.upsamplers.0.conv", blk["channels"], blk["channels"], 3)
    ini.norm("conv_norm_out", boc[0])
    ini.conv("conv_out", cfg["out_channels"], boc[0], 3)
    return ini.sd


def make_pose_guider_weights(cfg: dict = POSE_GUIDER_CFG, seed: int = 3) -> SD:
    """src/models/pose_guider.py:16-49 (conv_out is zero-init there; re-randomised here)."""
    ini = _Init(seed)
    boc = cfg["block_out_channels"]
    ini.conv("conv_in", boc[0], cfg["conditioning_channels"], 3)
    for i in range(len(boc) - 1):
        ini.conv(f"blocks.{2 * i}", boc[i], boc[i], 3)
        ini.conv(f"blocks.{2 * i + 1}", boc[i + 1], boc[i], 3)
    ini.conv("conv_out", cfg["conditioning_embedding_channels"], boc[-1], 3)
    return ini.sd


def make_camera_encoder_weights(cfg: dict = CAMERA_ENCODER_CFG, seed: int = 4) -> SD:
    """src/cameractrl/pose_adaptor.py:162-223 restricted to the geometry the inference yaml
    uses (one level, sk=True, in==out so no in_conv/skep; zero_conv re-randomised)."""
    ini = _Init(seed)
    chans = cfg["channels"]
    assert len(chans) == 1 and cfg["sk"], "only the inference_v2.yaml geometry is restated"
    c = chans[0]
    ini.conv("encoder_conv_in", c, cfg["cin"], 3)
    for j in range(cfg["nums_rb"]):
        p = f"encoder_down_conv_blocks.0.{j}"
        ini.conv(p + ".block1", c, c, 3)
        ini.conv(p + ".block2", c, c, cfg["ksize"])
        a = f"encoder_down_attention_blocks.0.{j}"
        for ai, _ in enumerate(cfg["attention_block_types"]):
            ini.attn(f"{a}.attention_blocks.{ai}", c)
            if cfg["temporal_position_encoding"]:
                ini.sd[f"{a}.attention_blocks.{ai}.pos_encoder.pe"] = sinusoidal_pe(
                    cfg["temporal_position_encoding_max_len"], c
                )
            ini.norm(f"{a}.norms.{ai}", c)
        ini.ff(a + ".ff", c)
        ini.norm(a + ".ff_norm", c)
    ini.conv("zero_conv_layers.0", c, c, 1, bias=False)
    return ini.sd


# --------------------------------------------------------------------------------------
# third-party (diffusers 0.24.0) semantics, restated -- SURVEY.md appendix C
# --------------------------------------------------------------------------------------


def timestep_embedding(t: Tensor, dim: int = 320) -> Tensor:
    """diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)
    (call site src/models/unet_3d.py:93, 461)."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = t.float()[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def linear(sd: SD, p: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def attention(sd: SD, p: str, x: Tensor, ctx: Optional[Tensor] = None, heads: int = 8) -> Tensor:
    """diffusers Attention + AttnProcessor2_0: to_q/k/v (no bias) -> SDPA (scale d^-0.5,
    no mask) -> to_out[0] (bias)."""
    ctx = x if ctx is None else ctx
    q, k, v = linear(sd, p + ".to_q", x), linear(sd, p + ".to_k", ctx), linear(sd, p + ".to_v", ctx)
    B, L, C = q.shape
    d = C // heads
    q = q.view(B, L, heads, d).transpose(1, 2)
    k = k.view(B, -1, heads, d).transpose(1, 2)
    v = v.view(B, -1, heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.transpose(1, 2).reshape(B, L, C)
    return linear(sd, p + ".to_out.0", o)


def feed_forward(sd: SD, p: str, x: Tensor) -> Tensor:
    """diffusers FeedForward(activation_fn="geglu"): Linear(C,8C) -> h*gelu(g) (erf) -> Linear(4C,C)."""
    h, g = linear(sd, p + ".net.0.proj", x).chunk(2, dim=-1)
    return linear(sd, p + ".net.2", h * F.gelu(g))


def layer_norm(sd: SD, p: str, x: Tensor) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def group_norm(sd: SD, p: str, x: Tensor, groups: int, eps: float) -> Tensor:
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


def conv2d(sd: SD, p: str, x: Tensor, stride: int = 1, padding: int = 1) -> Tensor:
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


# --------------------------------------------------------------------------------------
# blocks (all on the `(b f) c h w` / `(b f) n c` view the reference rearranges into)
# --------------------------------------------------------------------------------------


def resnet_block(sd: SD, p: str, x: Tensor, temb: Tensor, groups: int, eps: float) -> Tensor:
    """ResnetBlock3D.forward (src/models/resnet.py:215-245), per-frame GroupNorm
    (InflatedGroupNorm, resnet.py:18-26).  `temb` is already expanded per image [(b f), 1280]."""
    h = F.silu(group_norm(sd, p + ".norm1", x, groups, eps))
    h = conv2d(sd, p + ".conv1", h)
    h = h + linear(sd, p + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = F.silu(group_norm(sd, p + ".norm2", h, groups, eps))
    h = conv2d(sd, p + ".conv2", h)
    if (p + ".conv_shortcut.weight") in sd:
        x = conv2d(sd, p + ".conv_shortcut", x, padding=0)
    return x + h  # output_scale_factor == 1.0


def spatial_transformer(
    sd: SD,
    p: str,
    x: Tensor,
    ehs: Tensor,
    bank: Optional[Tensor],
    video_length: int,
    do_cfg: bool,
    heads: int,
    groups: int,
    bank_out: Optional[dict] = None,
) -> Tensor:
    """Transformer3DModel.forward (src/models/transformer_3d.py:103-169) with the read-mode
    patched block forward (src/models/mutual_self_attention.py:147-186, 187-228).

    x: [(b f), C, h, w]; ehs: [b, 1, 768]; bank: [b, Nb, C] (already fp16-rounded by
    `update`, mutual_self_attention.py:338) or None (no reference injection)."""
    n, C, hh, ww = x.shape
    res = x
    h = group_norm(sd, p + ".norm", x, groups, 1e-6)
    h = conv2d(sd, p + ".proj_in", h, padding=0)
    h = h.permute(0, 2, 3, 1).reshape(n, hh * ww, C)
    t = p + ".transformer_blocks.0"
    ehs_f = ehs.repeat_interleave(video_length, dim=0)  # 'b n c -> (b f) n c'
    nh = layer_norm(sd, t + ".norm1", h)
    if bank_out is not None:  # write mode: self.bank.append(norm_hidden_states.clone()), mutual_self_attention.py:137-138
        bank_out[p] = nh.clone()
    if bank is not None:
        bank_f = bank.to(nh.dtype).repeat_interleave(video_length, dim=0)
        kv = torch.cat([nh, bank_f], dim=1)
        out = attention(sd, t + ".attn1", nh, kv, heads) + h
        if do_cfg:
            half = n // 2  # first half of the batch is the unconditional one
            out_c = out.clone()
            out_c[:half] = attention(sd, t + ".attn1", nh[:half], nh[:half], heads) + h[:half]
            out = out_c
        h = out
    else:
        h = attention(sd, t + ".attn1", nh, None, heads) + h
    nh = layer_norm(sd, t + ".norm2", h)
    h = attention(sd, t + ".attn2", nh, ehs_f, heads) + h
    h = feed_forward(sd, t + ".ff", layer_norm(sd, t + ".norm3", h)) + h
    h = h.reshape(n, hh, ww, C).permute(0, 3, 1, 2)
    h = conv2d(sd, p + ".proj_out", h, padding=0)
    return h + res


def temporal_self_attention(sd: SD, p: str, x: Tensor, video_length: int, heads: int) -> Tensor:
    """VersatileAttention.forward (src/models/motion_module.py:351-388): `(b f) d c -> (b d) f c`,
    + sinusoidal PE, self-attention over f, back.  x is the *normed* hidden state."""
    n, N, C = x.shape
    b = n // video_length
    h = x.view(b, video_length, N, C).permute(0, 2, 1, 3).reshape(b * N, video_length, C)
    pe_key = p + ".pos_encoder.pe"
    if pe_key in sd:
        h = h + sd[pe_key][:, :video_length]
    h = attention(sd, p, h, None, heads)
    return h.view(b, N, video_length, C).permute(0, 2, 1, 3).reshape(n, N, C)


def temporal_transformer_block(sd: SD, b: str, h: Tensor, video_length: int, heads: int, n_attn: int) -> Tensor:
    """TemporalTransformerBlock.forward (src/models/motion_module.py:236-259; the cameractrl
    copy src/cameractrl/motion_module.py:288-299 is the same arithmetic)."""
    for ai in range(n_attn):
        nh = layer_norm(sd, f"{b}.norms.{ai}", h)
        h = temporal_self_attention(sd, f"{b}.attention_blocks.{ai}", nh, video_length, heads) + h
    return feed_forward(sd, b + ".ff", layer_norm(sd, b + ".ff_norm", h)) + h


def motion_module(sd: SD, p: str, x: Tensor, video_length: int, mmk: dict, groups: int) -> Tensor:
    """VanillaTemporalModule / TemporalTransformer3DModel.forward (motion_module.py:77-91,146-182)."""
    n, C, hh, ww = x.shape
    t = p + ".temporal_transformer"
    res = x
    h = group_norm(sd, t + ".norm", x, groups, 1e-6)
    h = h.permute(0, 2, 3, 1).reshape(n, hh * ww, C)
    h = linear(sd, t + ".proj_in", h)
    for li in range(mmk.get("num_transformer_block", 1)):
        h = temporal_transformer_block(
            sd, f"{t}.transformer_blocks.{li}", h, video_length, mmk["num_attention_heads"],
            len(mmk["attention_block_types"]),
        )
    h = linear(sd, t + ".proj_out", h)
    h = h.reshape(n, hh, ww, C).permute(0, 3, 1, 2)
    return h + res


def unet3d_forward(
    sd: SD,
    cfg: dict,
    sample: Tensor,
    timestep,
    encoder_hidden_states: Tensor,
    pose_cond_fea: Optional[Tensor] = None,
    banks: Optional[Dict[str, Tensor]] = None,
    do_cfg: bool = True,
    taps: Optional[dict] = None,
) -> Tensor:
    """UNet3DConditionModel.forward (src/models/unet_3d.py:397-577) in read mode.

    sample [b,4,f,h,w]; encoder_hidden_states [b,1,768]; pose_cond_fea [b,320,f,h,w];
    banks: {transformer location prefix -> [b, Nb, C]} (see transformer_locations)."""
    spec = unet3d_spec(cfg)
    b, _, f, hh, ww = sample.shape
    groups, eps, heads = cfg["norm_num_groups"], cfg["norm_eps"], cfg["attention_head_dim"]
    mmk = cfg.get("motion_module_kwargs", {})
    banks = banks or {}

    def to2d(x):  # 'b c f h w -> (b f) c h w'
        return x.permute(0, 2, 1, 3, 4).reshape(b * f, x.shape[1], x.shape[3], x.shape[4])

    t = torch.as_tensor(timestep)
    t = t[None] if t.ndim == 0 else t
    t = t.expand(b)
    emb = timestep_embedding(t, spec["boc"][0])
    emb = linear(sd, "time_embedding.linear_2", F.silu(linear(sd, "time_embedding.linear_1", emb)))
    emb_f = emb.repeat_interleave(f, dim=0)  # per image

    x = conv2d(sd, "conv_in", to2d(sample))
    if pose_cond_fea is not None:
        x = x + to2d(pose_cond_fea)
    skips = [x]

    def tf(p, x):
        return spatial_transformer(sd, p, x, encoder_hidden_states, banks.get(p), f, do_cfg, heads, groups)

    def tap(name, v):
        if taps is not None:
            taps[name] = v.detach().clone()

    for blk in spec["down"]:
        p = blk["prefix"]
        for j in range(len(blk["resnets"])):
            x = resnet_block(sd, f"{p}.resnets.{j}", x, emb_f, groups, eps)
            tap(f"{p}.resnets.{j}", x)
            if blk["attn"]:
                x = tf(f"{p}.attentions.{j}", x)
                tap(f"{p}.attentions.{j}", x)
            if blk["motion"]:
                x = motion_module(sd, f"{p}.motion_modules.{j}", x, f, mmk, groups)
                tap(f"{p}.motion_modules.{j}", x)
            skips.append(x)
        if blk["downsample"]:
            x = conv2d(sd, f"{p}.downsamplers.0.conv", x, stride=2, padding=1)
            skips.append(x)
    x = resnet_block(sd, "mid_block.resnets.0", x, emb_f, groups, eps)
    x = tf("mid_block.attentions.0", x)
    if spec["mid"]["motion"]:
        x = motion_module(sd, "mid_block.motion_modules.0", x, f, mmk, groups)
    x = resnet_block(sd, "mid_block.resnets.1", x, emb_f, groups, eps)
    tap("mid_block", x)
    for blk in spec["up"]:
        p = blk["prefix"]
        for j in range(len(blk["resnets"])):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet_block(sd, f"{p}.resnets.{j}", x, emb_f, groups, eps)
            if blk["attn"]:
                x = tf(f"{p}.attentions.{j}", x)
            if blk["motion"]:
                x = motion_module(sd, f"{p}.motion_modules.{j}", x, f, mmk, groups)
            tap(f"{p}.{j}", x)
        if blk["upsample"]:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = conv2d(sd, f"{p}.upsamplers.0.conv", x)
    x = F.silu(group_norm(sd, "conv_norm_out", x, groups, eps))
    x = conv2d(sd, "conv_out", x)
    return x.view(b, f, -1, hh, ww).permute(0, 2, 1, 3, 4)


def reference_net_cfg(cfg3d: dict) -> dict:
    """The ReferenceNet is the same SD-1.5 geometry without motion modules."""
    c = dict(cfg3d)
    c.update(use_motion_module=False)
    return c


def make_reference_net_weights(cfg: dict, seed: int = 5) -> SD:
    sd = make_unet3d_weights(reference_net_cfg(cfg), seed)
    for k in ("conv_norm_out.weight", "conv_norm_out.bias", "conv_out.weight", "conv_out.bias"):
        sd.pop(k)  # removed in the reference's ReferenceNet (src/models/unet_2d_condition.py:645-653)
    return sd


def reference_net_banks(sd: SD, cfg: dict, ref_latents: Tensor, encoder_hidden_states: Tensor) -> Dict[str, Tensor]:
    """ReferenceNet write pass (src/pipelines/pipeline_pose2vid_long.py:470-480 ->
    src/models/unet_2d_condition.py:872-1308 with mutual_self_attention.py:137-146): an SD UNet on
    the reference latent at t = 0; returns {transformer location -> norm1 features [b, N, C]}.
    NOTE: restated from the 3-D blocks with one frame (diffusers' ResnetBlock2D / Transformer2DModel
    are not available here to pin against; SURVEY.md appendix C last bullet)."""
    cfg = reference_net_cfg(cfg)
    spec = unet3d_spec(cfg)
    b = ref_latents.shape[0]
    groups, eps, heads = cfg["norm_num_groups"], cfg["norm_eps"], cfg["attention_head_dim"]
    banks: Dict[str, Tensor] = {}
    emb = timestep_embedding(torch.zeros(b), spec["boc"][0])
    emb = linear(sd, "time_embedding.linear_2", F.silu(linear(sd, "time_embedding.linear_1", emb)))
    x = conv2d(sd, "conv_in", ref_latents)
    skips = [x]

    def tf(p, x):
        return spatial_transformer(sd, p, x, encoder_hidden_states, None, 1, False, heads, groups, bank_out=banks)

    for blk in spec["down"]:
        p = blk["prefix"]
        for j in range(len(blk["resnets"])):
            x = resnet_block(sd, f"{p}.resnets.{j}", x, emb, groups, eps)
            if blk["attn"]:
                x = tf(f"{p}.attentions.{j}", x)
            skips.append(x)
        if blk["downsample"]:
            x = conv2d(sd, f"{p}.downsamplers.0.conv", x, stride=2, padding=1)
            skips.append(x)
    x = resnet_block(sd, "mid_block.resnets.0", x, emb, groups, eps)
    x = tf("mid_block.attentions.0", x)
    x = resnet_block(sd, "mid_block.resnets.1", x, emb, groups, eps)
    for blk in spec["up"]:
        p = blk["prefix"]
        for j in range(len(blk["resnets"])):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet_block(sd, f"{p}.resnets.{j}", x, emb, groups, eps)
            if blk["attn"]:
                x = tf(f"{p}.attentions.{j}", x)
        if blk["upsample"]:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = conv2d(sd, f"{p}.upsamplers.0.conv", x)
    return banks


# --------------------------------------------------------------------------------------
# conditioning encoders
# --------------------------------------------------------------------------------------


def pose_guider_forward(sd: SD, cond: Tensor, cfg: dict = POSE_GUIDER_CFG) -> Tensor:
    """PoseGuider.forward (src/models/pose_guider.py:51-61). cond [b,3,f,H,W] -> [b,320,f,H/8,W/8]."""
    b, _, f, H, W = cond.shape
    x = cond.permute(0, 2, 1, 3, 4).reshape(b * f, -1, H, W)
    x = F.silu(conv2d(sd, "conv_in", x))
    nb = 2 * (len(cfg["block_out_channels"]) - 1)
    for i in range(nb):
        x = F.silu(conv2d(sd, f"blocks.{i}", x, stride=2 if i % 2 == 1 else 1))
    x = conv2d(sd, "conv_out", x)
    return x.view(b, f, x.shape[1], x.shape[2], x.shape[3]).permute(0, 2, 1, 3, 4)


def camera_encoder_forward(sd: SD, x: Tensor, cfg: dict = CAMERA_ENCODER_CFG) -> Tensor:
    """CameraPoseEncoder.forward (src/cameractrl/pose_adaptor.py:232-248), first (only) feature.
    x [b,6,f,H,W] -> [(b f), 320, H/8, W/8]."""
    b, _, f, H, W = x.shape
    x = x.permute(0, 2, 1, 3, 4).reshape(b * f, -1, H, W)
    x = F.pixel_unshuffle(x, cfg["downscale_factor"])
    x = conv2d(sd, "encoder_conv_in", x)
    heads = cfg["temporal_attention_nhead"]
    for j in range(cfg["nums_rb"]):
        p = f"encoder_down_conv_blocks.0.{j}"
        h = conv2d(sd, p + ".block1", x)
        h = F.relu(h)
        h = conv2d(sd, p + ".block2", h, padding=cfg["ksize"] // 2)
        x = h + x
        n, C, hh, ww = x.shape
        tok = x.permute(0, 2, 3, 1).reshape(n, hh * ww, C)  # same token view as the motion module
        tok = temporal_transformer_block(
            sd, f"encoder_down_attention_blocks.0.{j}", tok, f, heads, len(cfg["attention_block_types"])
        )
        x = tok.reshape(n, hh, ww, C).permute(0, 3, 1, 2)
    return conv2d(sd, "zero_conv_layers.0", x, padding=0)


# --------------------------------------------------------------------------------------
# scheduler, windows, pipeline loop
# --------------------------------------------------------------------------------------


class DDIM:
    """diffusers DDIMScheduler with configs/inference/inference_v2.yaml:24-33
    (linear betas, steps_offset 1, clip_sample False, v_prediction, zero-terminal-SNR,
    trailing spacing, set_alpha_to_one) -- SURVEY.md appendix C."""

    def __init__(self, beta_start=0.00085, beta_end=0.012, num_train_timesteps=1000):
        betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        alphas = 1.0 - betas
        abar = torch.cumprod(alphas, dim=0)
        s = abar.sqrt()
        s0, sT = s[0].clone(), s[-1].clone()
        s = (s - sT) * s0 / (s0 - sT)
        abar = s**2
        alphas = torch.cat([abar[0:1], abar[1:] / abar[:-1]])
        self.betas = 1 - alphas
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0)
        self.num_train_timesteps = num_train_timesteps
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n: int):
        self.num_inference_steps = n
        ts = np.round(np.arange(self.num_train_timesteps, 0, -self.num_train_timesteps / n)) - 1
        self.timesteps = torch.from_numpy(ts.astype(np.int64))

    def coeffs(self, t: int):
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a = self.alphas_cumprod[t]
        ap = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        return a, ap

    def step(self, v: Tensor, t: int, x: Tensor) -> Tensor:
        a, ap = self.coeffs(int(t))
        x0 = a.sqrt() * x - (1 - a).sqrt() * v
        eps = a.sqrt() * v + (1 - a).sqrt() * x
        return ap.sqrt() * x0 + (1 - ap).sqrt() * eps


def ordered_halving(val: int) -> float:
    """src/pipelines/context.py:7-12."""
    return int(f"{val:064b}"[::-1], 2) / (1 << 64)


def uniform_windows(step, num_steps, num_frames, context_size, context_stride=3, context_overlap=4, closed_loop=True):
    """src/pipelines/context.py:15-42."""
    if num_frames <= context_size:
        return [list(range(num_frames))]
    out = []
    context_stride = min(context_stride, int(np.ceil(np.log2(num_frames / context_size))) + 1)
    for context_step in 1 << np.arange(context_stride):
        pad = int(round(num_frames * ordered_halving(step)))
        for j in range(
            int(ordered_halving(step) * context_step) + pad,
            num_frames + pad + (0 if closed_loop else -context_overlap),
            (context_size * context_step - context_overlap),
        ):
            out.append([e % num_frames for e in range(j, j + context_size * context_step, context_step)])
    return out


def denoise_loop(
    unet_sd: SD,
    cfg: dict,
    pg_sd: SD,
    cam_sd: SD,
    latents: Tensor,
    pose_cond: Tensor,
    camera_embedding: Tensor,
    clip_embeds: Tensor,
    banks: Dict[str, Tensor],
    num_inference_steps: int,
    guidance_scale: float,
    context_frames: int = 24,
    context_stride: int = 1,
    context_overlap: int = 4,
    max_steps: Optional[int] = None,
    trace: Optional[list] = None,
) -> Tensor:
    """The denoising loop body of Pose2VideoPipeline.__call__
    (src/pipelines/pipeline_pose2vid_long.py:454-571).  latents [1,4,F,h,w]; pose_cond
    [1,3,F,H,W] in [0,1]; camera_embedding [1,6,F,H,W]; clip_embeds [1,1,768] (the conditional
    embedding; the unconditional one is zeros, :387); banks from the ReferenceNet write pass."""
    do_cfg = guidance_scale > 1.0
    sched = DDIM()
    sched.set_timesteps(num_inference_steps)
    ehs = torch.cat([torch.zeros_like(clip_embeds), clip_embeds], dim=0) if do_cfg else clip_embeds
    rep = 2 if do_cfg else 1
    F_ = latents.shape[2]
    for i, t in enumerate(sched.timesteps.tolist()):
        if max_steps is not None and i >= max_steps:
            break
        noise_pred = torch.zeros(latents.shape[0] * rep, *latents.shape[1:])
        counter = torch.zeros(1, 1, F_, 1, 1)
        windows = uniform_windows(0, num_inference_steps, F_, context_frames, context_stride, context_overlap)
        for c in windows:
            lat_in = latents[:, :, c].repeat(rep, 1, 1, 1, 1)
            pose_fea = pose_guider_forward(pg_sd, pose_cond[:, :, c])
            cam = camera_encoder_forward(cam_sd, camera_embedding[:, :, c])
            cam = cam.view(1, len(c), *cam.shape[1:]).permute(0, 2, 1, 3, 4)
            cond = pose_fea.repeat(rep, 1, 1, 1, 1) + cam.repeat(rep, 1, 1, 1, 1)
            pred = unet3d_forward(unet_sd, cfg, lat_in, t, ehs[: lat_in.shape[0]], cond, banks, do_cfg)
            noise_pred[:, :, c] = noise_pred[:, :, c] + pred
            counter[:, :, c] = counter[:, :, c] + 1
        if do_cfg:
            u, c_ = (noise_pred / counter).chunk(2)
            noise_pred = u + guidance_scale * (c_ - u)
        else:
            noise_pred = noise_pred / counter
        latents = sched.step(noise_pred, t, latents)
        if trace is not None:
            trace.append(latents.clone())
    return latents


# --------------------------------------------------------------------------------------
# camera front-end
# --------------------------------------------------------------------------------------


def quaternion_to_rotation_matrix(qx, qy, qz, qw):
    """src/dataset/dance_image_h_v_camera.py:66-76."""
    return np.array(
        [
            [1 - 2 * qy**2 - 2 * qz**2, 2 * qx * qy - 2 * qz * qw, 2 * qx * qz + 2 * qy * qw],
            [2 * qx * qy + 2 * qz * qw, 1 - 2 * qx**2 - 2 * qz**2, 2 * qy * qz - 2 * qx * qw],
            [2 * qx * qz - 2 * qy * qw, 2 * qy * qz + 2 * qx * qw, 1 - 2 * qx**2 - 2 * qy**2],
        ]
    )


def camera_c2w_and_intrinsics(entry: Sequence[float], image_scale):
    """Camera.__init__ for the c2w-convention datasets ("test"/"pexels"/... branch),
    src/dataset/dance_image_h_v_camera.py:17-64. Returns (fx, fy, cx, cy, c2w, w2c)."""
    assert len(entry) in (10, 11)
    if image_scale[0] > image_scale[1]:
        fx = entry[8]
        fy = fx * (image_scale[0] / image_scale[1])
    else:
        fy = entry[9]
        fx = fy * (image_scale[1] / image_scale[0])
    tx, ty, tz = entry[1:4]
    q = np.array(entry[4:8], dtype=np.float64)
    scale = entry[10] if len(entry) == 11 else 1.0
    q = q / np.linalg.norm(q)
    c2w = np.eye(4)
    c2w[:3, :3] = quaternion_to_rotation_matrix(*q)
    c2w[:3, 3] = np.array([tx, ty, tz]) * scale
    return fx, fy, 0.5, 0.5, c2w, np.linalg.inv(c2w)


def plucker_from_entries(entries: Sequence[Sequence[float]], img_size) -> Tensor:
    """scripts/pose2vid.py:53-83 (camera_file_to_embedding) with is_same_video semantics applied
    by the caller: entries[0] is the reference camera, entries[1:] the target frames.
    img_size = (W, H).  Returns [1, F, 6, H, W]."""
    cams = [camera_c2w_and_intrinsics(e, img_size) for e in entries]
    K = np.asarray(
        [[c[0] * img_size[0], c[1] * img_size[1], c[2] * img_size[0], c[3] * img_size[1]] for c in cams[1:]],
        dtype=np.float32,
    )
    abs2rel = np.eye(4) @ cams[0][5]  # get_relative_pose, dance_image_h_v_camera.py:29-43 (scripts copy)
    poses = [np.eye(4)] + [abs2rel @ c[4] for c in cams[1:]]
    c2w = torch.as_tensor(np.array(poses, dtype=np.float32)[1:])[None]
    K = torch.as_tensor(K)[None]
    pl = ray_condition(K, c2w, img_size[1], img_size[0])
    return pl[0].permute(0, 3, 1, 2).contiguous()[None]


def ray_condition(K: Tensor, c2w: Tensor, H: int, W: int) -> Tensor:
    """src/dataset/dance_image_h_v_camera.py:88-130 (flip_flag=None). -> [B, V, H, W, 6]."""
    B, V = K.shape[:2]
    j, i = torch.meshgrid(
        torch.linspace(0, H - 1, H, dtype=c2w.dtype), torch.linspace(0, W - 1, W, dtype=c2w.dtype), indexing="ij"
    )
    i = i.reshape(1, 1, H * W).expand(B, V, H * W) + 0.5
    j = j.reshape(1, 1, H * W).expand(B, V, H * W) + 0.5
    fx, fy, cx, cy = K.chunk(4, dim=-1)
    zs = torch.ones_like(i)
    xs = (i - cx) / fx * zs
    ys = (j - cy) / fy * zs
    d = torch.stack((xs, ys, zs), dim=-1)
    d = d / d.norm(dim=-1, keepdim=True)
    rays_d = d @ c2w[..., :3, :3].transpose(-1, -2)
    rays_o = c2w[..., :3, 3][:, :, None].expand_as(rays_d)
    rays_dxo = torch.cross(rays_o, rays_d, dim=-1)
    return torch.cat([rays_dxo, rays_d], dim=-1).reshape(B, V, H, W, 6)


# ---------------------------------------------------------------------------------------------------------------
# Latent interpolation between denoised frames (Pose2VideoPipeline.interpolate_latents,
# /root/reference/src/pipelines/pipeline_pose2vid_long.py:294-337; blending functions /root/reference/src/pipelines/utils.py:15-30).
# Pinned by oracle/gen_interp_golden.py (runs the reference's own function bodies) -> tests/golden/latent_interp.npz.
def interp_linear(v0: Tensor, v1: Tensor, t: float) -> Tensor:
    return (1.0 - t) * v0 + t * v1  # utils.py:15-16


def interp_slerp(v0: Tensor, v1: Tensor, t: float, dot_threshold: float = 0.9995) -> Tensor:
    """utils.py:19-30: the angle is that of the two WHOLE tensors (one scalar); nearly parallel -> linear blend"""
    dot = ((v0 / v0.norm()) * (v1 / v1.norm())).sum()
    if dot.abs() > dot_threshold:
        return (1.0 - t) * v0 + t * v1
    omega = dot.acos()
    return (((1.0 - t) * omega).sin() * v0 + (t * omega).sin() * v1) / omega.sin()


def interpolate_latents(latents: Tensor, factor: int, slerp: bool) -> Tensor:
    """pipeline_pose2vid_long.py:294-337: [B,C,F,h,w] -> [B,C,(F-1)*factor+1,h,w]; between frames i and i+1 the blends at
    t = 1/factor .. (factor-1)/factor; factor < 2 returns the input."""
    if factor < 2:
        return latents
    blend = interp_slerp if slerp else interp_linear
    F = latents.shape[2]
    frames = []
    for i in range(F - 1):
        v0, v1 = latents[:, :, i], latents[:, :, i + 1]
        frames.append(v0)
        for j in range(1, factor):
            frames.append(blend(v0, v1, j / factor))
    frames.append(latents[:, :, F - 1])
    return torch.stack(frames, dim=2)
