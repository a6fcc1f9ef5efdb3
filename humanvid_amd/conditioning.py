"""PoseGuider and CameraPoseEncoder: drop-in parameter containers + native forwards.

 * PoseGuider   -- /root/reference/src/models/pose_guider.py:16-61: 8 3x3 convs (3->16->16->32(s2)
   ->32->96(s2)->96->256(s2)->320) with SiLU, executed as hv_conv3x3 launches with the activation in
   the epilogue; narrow channel counts are zero-padded to the kernel's 32-channel granule at pack
   time (padded output channels are exactly zero, so the next conv ignores them).
 * CameraPoseEncoder -- /root/reference/src/cameractrl/pose_adaptor.py:160-248 with the geometry of
   configs/inference/inference_v2.yaml:38-50: PixelUnshuffle(8) -> conv3x3 384->320 -> 2 x
   [ResnetBlock(3x3, ReLU, 1x1, +x) ; temporal transformer block (LN -> PE -> self-attention over
   frames -> +res ; LN -> GEGLU FF -> +res)] -> 1x1 zero-conv.

Both are timestep independent; the pipeline evaluates them once per clip and window instead of
once per step (the reference recomputes them every step, pipeline_pose2vid_long.py:526-539).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from . import _abi as A
from . import lib as hvlib
from . import ops, packing
from .runner import Runner, Workspace
from .unet3d import AttentionParams, FeedForwardParams, InflatedConv3d, PositionalEncoding, _Holder, zero_module

BF16 = torch.bfloat16
F32 = torch.float32


def _pack_conv_padded(w: torch.Tensor, b: Optional[torch.Tensor], device):
    """[Cout, Cin, 3, 3] -> packed [Cout_pad, 9, Cin_pad] with both channel counts rounded up to 32."""
    cout, cin = w.shape[:2]
    cop, cip = packing.round_up(cout, 32), packing.round_up(cin, 32)
    wp = torch.zeros(cop, 9, cip, dtype=BF16)
    wp[:cout] = packing.pack_conv3x3(w.float().cpu(), cip)
    bp = torch.zeros(cop, dtype=F32)
    if b is not None:
        bp[:cout] = b.float().cpu()
    return wp.to(device), bp.to(device)


class PoseGuider(nn.Module):
    def __init__(self, conditioning_embedding_channels: int, conditioning_channels: int = 3,
                 block_out_channels: Tuple[int] = (16, 32, 64, 128)):
        super().__init__()
        self.conv_in = InflatedConv3d(conditioning_channels, block_out_channels[0], 3, padding=1)
        self.blocks = nn.ModuleList([])
        for i in range(len(block_out_channels) - 1):
            cin, cout = block_out_channels[i], block_out_channels[i + 1]
            self.blocks.append(InflatedConv3d(cin, cin, 3, padding=1))
            self.blocks.append(InflatedConv3d(cin, cout, 3, padding=1, stride=2))
        self.conv_out = zero_module(InflatedConv3d(block_out_channels[-1], conditioning_embedding_channels, 3, padding=1))
        self._packed = None
        self._ws = None

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = None
        return super().load_state_dict(*a, **k)

    def _layers(self, device):
        if self._packed is None:
            convs = [self.conv_in] + list(self.blocks) + [self.conv_out]
            self._packed = [(*_pack_conv_padded(c.weight.detach(), c.bias.detach(), device), c.stride[0], c.out_channels)
                            for c in convs]
            self._ws = Workspace(device)
        return self._packed

    @torch.no_grad()
    def forward_nhwc(self, cond: torch.Tensor) -> torch.Tensor:
        """cond [b,3,f,H,W] (cuda, fp32/bf16) -> [(b f), H/8, W/8, C_out] bf16 channels-last."""
        dev = hvlib.require_gpu()
        if cond.device.type != "cuda":
            raise RuntimeError("PoseGuider needs CUDA/HIP tensors: there is no CPU path")
        L, st = hvlib.load(), hvlib.current_stream()
        layers = self._layers(dev)
        b, c, f, H, W = cond.shape
        src = cond if cond.dtype in (F32, BF16) else cond.float()
        x = self._ws.get("in", (b * f, H, W, 32))
        ops.pack_ncfhw(L, st, src.contiguous(), x)
        for i, (wp, bp, stride, cout) in enumerate(layers):
            last = i == len(layers) - 1
            if stride == 2:
                H, W = (H + 1) // 2, (W + 1) // 2
            y = self._ws.get(f"a{i}", (b * f, H, W, wp.shape[0]))
            ops.conv3x3(L, st, x, wp, y, mode=A.CONV_S2 if stride == 2 else A.CONV_S1, bias=bp,
                        out_act=A.ACT_NONE if last else A.ACT_SILU)
            x = y
        return x[..., : layers[-1][3]] if x.shape[-1] != layers[-1][3] else x

    @torch.no_grad()
    def forward(self, conditioning):
        b, c, f, H, W = conditioning.shape
        y = self.forward_nhwc(conditioning).contiguous()
        out = torch.empty(b, y.shape[-1], f, y.shape[1], y.shape[2], dtype=F32, device=y.device)
        ops.unpack_nhwc(hvlib.load(), hvlib.current_stream(), y, out)
        return out.to(conditioning.dtype if conditioning.dtype.is_floating_point else F32)


# ------------------------------------------------------------------------------------------- camera
class _ResnetBlock(_Holder):  # src/cameractrl/pose_adaptor.py:102-135 (sk=True, in==out geometry)
    def __init__(self, in_c, out_c, down, ksize=3, sk=False, use_conv=True):
        super().__init__()
        if down or in_c != out_c or not sk:
            raise NotImplementedError("only the inference_v2.yaml camera-encoder geometry is supported "
                                      "(sk=True, equal widths, no down-sampling)")
        ps = ksize // 2
        self.in_conv = None
        self.block1 = nn.Conv2d(out_c, out_c, 3, 1, 1)
        self.act = nn.ReLU()
        self.block2 = nn.Conv2d(out_c, out_c, ksize, 1, ps)
        self.skep = None
        self.down = down


class _TemporalSelfAttention(AttentionParams):  # src/cameractrl/motion_module.py:323-338
    def __init__(self, dim, heads, dim_head, temporal_position_encoding, max_len):
        super().__init__(dim, None, heads, dim_head)
        self.pos_encoder = PositionalEncoding(dim, max_len) if temporal_position_encoding else None


class _CamTemporalTransformerBlock(_Holder):  # src/cameractrl/motion_module.py:236-286
    def __init__(self, dim, heads, dim_head, attention_block_types, temporal_position_encoding, max_len):
        super().__init__()
        for name in attention_block_types:
            if name != "Temporal_Self":
                raise NotImplementedError(name)
        self.attention_blocks = nn.ModuleList(
            [_TemporalSelfAttention(dim, heads, dim_head, temporal_position_encoding, max_len)
             for _ in attention_block_types])
        self.norms = nn.ModuleList([nn.LayerNorm(dim) for _ in attention_block_types])
        self.ff = FeedForwardParams(dim)
        self.ff_norm = nn.LayerNorm(dim)


class CameraPoseEncoder(nn.Module):
    def __init__(self, downscale_factor, channels=[320, 640, 1280, 1280], nums_rb=3, cin=64, ksize=3, sk=False,
                 use_conv=True, compression_factor=1, temporal_attention_nhead=8,
                 attention_block_types=("Temporal_Self",), temporal_position_encoding=False,
                 temporal_position_encoding_max_len=16, rescale_output_factor=1.0):
        super().__init__()
        if len(channels) != 1 or compression_factor != 1 or rescale_output_factor != 1.0:
            raise NotImplementedError("only the inference_v2.yaml camera-encoder geometry is supported")
        self.downscale_factor = downscale_factor
        self.unshuffle = nn.PixelUnshuffle(downscale_factor)
        self.channels, self.nums_rb, self.ksize = list(channels), nums_rb, ksize
        self.attention_block_types = tuple(attention_block_types)
        self.temporal_attention_nhead = temporal_attention_nhead
        c = channels[0]
        self.encoder_down_conv_blocks = nn.ModuleList([nn.ModuleList(
            [_ResnetBlock(c, c, down=False, ksize=ksize, sk=sk, use_conv=use_conv) for _ in range(nums_rb)])])
        self.encoder_down_attention_blocks = nn.ModuleList([nn.ModuleList(
            [_CamTemporalTransformerBlock(c, temporal_attention_nhead, c // temporal_attention_nhead,
                                          self.attention_block_types, temporal_position_encoding,
                                          temporal_position_encoding_max_len) for _ in range(nums_rb)])])
        zc = nn.Conv2d(c, c, kernel_size=1, stride=1, padding=0, bias=False)
        nn.init.zeros_(zc.weight)
        self.zero_conv_layers = nn.ModuleList([zc])
        self.encoder_conv_in = nn.Conv2d(cin, c, 3, 1, 1)
        self._run: Optional[Runner] = None

    @property
    def dtype(self) -> torch.dtype:
        return next(self.parameters()).dtype

    def _apply(self, fn, *a, **k):
        self._run = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._run = None
        return super().load_state_dict(*a, **k)

    def _runner(self, device) -> Runner:
        if self._run is not None:
            return self._run
        sd = self.state_dict()
        w: Dict[str, torch.Tensor] = {}

        def d(t, dtype=None):
            return t.detach().to(device=device, dtype=dtype or t.dtype).contiguous()

        w["conv_in.w"] = d(packing.pack_conv3x3(sd["encoder_conv_in.weight"].float()))
        w["conv_in.bias"] = d(sd["encoder_conv_in.bias"], F32)
        for j in range(self.nums_rb):
            p = f"encoder_down_conv_blocks.0.{j}"
            w[p + ".block1.w"] = d(packing.pack_conv3x3(sd[p + ".block1.weight"].float()))
            w[p + ".block1.bias"] = d(sd[p + ".block1.bias"], F32)
            if self.ksize == 1:
                w[p + ".block2.w"] = d(packing.pack_linear(sd[p + ".block2.weight"].float()))
            else:
                w[p + ".block2.w"] = d(packing.pack_conv3x3(sd[p + ".block2.weight"].float()))
            w[p + ".block2.bias"] = d(sd[p + ".block2.bias"], F32)
            a = f"encoder_down_attention_blocks.0.{j}"
            for ai in range(len(self.attention_block_types)):
                ab = f"{a}.attention_blocks.{ai}"
                qkv = torch.cat([sd[ab + f".to_{x}.weight"] for x in "qkv"], dim=0).float()
                gamma, beta = sd[f"{a}.norms.{ai}.weight"].float(), sd[f"{a}.norms.{ai}.bias"].float()
                wf, cs, bf = packing.fold_layernorm(qkv, None, gamma, beta)
                w[ab + ".qkv.w"], w[ab + ".qkv.colsum"], w[ab + ".qkv.bias"] = d(wf), d(cs, F32), d(bf, F32)
                pe = sd.get(ab + ".pos_encoder.pe")
                if pe is not None:
                    w[ab + ".qkv.pe"] = d(packing.pe_table(pe[0].float(), qkv), F32)
                w[ab + ".to_out.0.w"] = d(packing.pack_linear(sd[ab + ".to_out.0.weight"].float()))
                w[ab + ".to_out.0.bias"] = d(sd[ab + ".to_out.0.bias"], F32)
            wf, cs, bf = packing.fold_layernorm(sd[a + ".ff.net.0.proj.weight"].float(), sd[a + ".ff.net.0.proj.bias"].float(),
                                                sd[a + ".ff_norm.weight"].float(), sd[a + ".ff_norm.bias"].float())
            order = packing.geglu_row_order(wf.shape[0])
            w[a + ".ff1.w"], w[a + ".ff1.colsum"], w[a + ".ff1.bias"] = d(wf[order]), d(cs[order], F32), d(bf[order], F32)
            w[a + ".ff.net.2.w"] = d(packing.pack_linear(sd[a + ".ff.net.2.weight"].float()))
            w[a + ".ff.net.2.bias"] = d(sd[a + ".ff.net.2.bias"], F32)
        w["zero_conv.w"] = d(packing.pack_linear(sd["zero_conv_layers.0.weight"].float()))
        self._run = Runner(device, w, Workspace(device), temporal_heads=self.temporal_attention_nhead)
        return self._run

    @torch.no_grad()
    def forward_nhwc_from_cameras(self, K: torch.Tensor, c2w: torch.Tensor, H: int, W: int,
                                  add: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Camera front-end on the device: K [f,4] = (fx, fy, cx, cy) in pixels and the
        relative camera-to-world matrices c2w [f,4,4] (what `ray_condition` consumes, dance_image_h_v_camera.py:88-130)
        -> the same feature as forward_nhwc(ray_condition(K, c2w, H, W) laid out [1,6,f,H,W]).  The Pluecker map is
        generated inside the PixelUnshuffle kernel and never stored."""
        dev = hvlib.require_gpu()
        run = self._runner(dev)
        f = K.shape[0]
        r = self.downscale_factor
        xs = run.ws.get("unshuffled", (f, H // r, W // r, 6 * r * r))
        ops.plucker_unshuffle(run.lib, run.st, K.to(dev, F32).contiguous(), c2w.to(dev, F32).contiguous(), H, W, r, xs)
        return self._encode(run, xs, 1, f, H // r, W // r, add)

    @torch.no_grad()
    def forward_nhwc(self, x: torch.Tensor, add: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x [b,6,f,H,W] fp32 (cuda) -> feature [(b f), H/8, W/8, C] bf16; `add` (same shape, e.g.
        the PoseGuider output) is summed in the zero-conv epilogue (pipeline_pose2vid_long.py:546)."""
        dev = hvlib.require_gpu()
        if x.device.type != "cuda":
            raise RuntimeError("CameraPoseEncoder needs CUDA/HIP tensors: there is no CPU path")
        run = self._runner(dev)
        b, c, f, H, W = x.shape
        r = self.downscale_factor
        xs = run.ws.get("unshuffled", (b * f, H // r, W // r, c * r * r))
        ops.pixel_unshuffle(run.lib, run.st, x.float().contiguous(), xs, r)
        return self._encode(run, xs, b, f, H // r, W // r, add)

    def _encode(self, run, xs, b, f, h, ww, add):
        L, st, w, ws = run.lib, run.st, run.w, run.ws
        n, N, C = b * f, h * ww, self.channels[0]
        cur = ws.get("x0", (n, h, ww, C))
        ops.conv3x3(L, st, xs, w["conv_in.w"], cur, bias=w["conv_in.bias"])
        for j in range(self.nums_rb):
            p = f"encoder_down_conv_blocks.0.{j}"
            h1 = ws.get("h1", (n, h, ww, C))
            ops.conv3x3(L, st, cur, w[p + ".block1.w"], h1, bias=w[p + ".block1.bias"], out_act=A.ACT_RELU)
            nxt = ws.get(f"x{j + 1}", (n, h, ww, C))
            if self.ksize == 1:
                ops.gemm(L, st, h1.view(n * N, C), w[p + ".block2.w"], nxt.view(n * N, C), bias=w[p + ".block2.bias"],
                         residual=cur.view(n * N, C))
            else:
                ops.conv3x3(L, st, h1, w[p + ".block2.w"], nxt, bias=w[p + ".block2.bias"], residual=cur)
            cur = nxt
            hid = cur.view(n * N, C)  # token-major view == '(b f) c h w -> (b h w) f c' by addressing
            a = f"encoder_down_attention_blocks.0.{j}"
            for ai in range(len(self.attention_block_types)):
                run.temporal_attention_block(f"{a}.attention_blocks.{ai}", hid, b, f, N, sharded=False)
            run.feed_forward(a + ".ff1", a + ".ff.net.2", hid)
        out = ws.get("feature", (n, h, ww, C))
        ops.gemm(L, st, cur.view(n * N, C), w["zero_conv.w"], out.view(n * N, C),
                 residual=None if add is None else add.reshape(n * N, C))
        return out

    @torch.no_grad()
    def forward(self, x) -> List[torch.Tensor]:
        b, c, f, H, W = x.shape
        y = self.forward_nhwc(x)
        out = torch.empty(b, y.shape[-1], f, y.shape[1], y.shape[2], dtype=F32, device=y.device)
        ops.unpack_nhwc(hvlib.load(), hvlib.current_stream(), y, out)
        out = out.permute(0, 2, 1, 3, 4).reshape(b * f, y.shape[-1], y.shape[1], y.shape[2])
        return [out.to(x.dtype if x.dtype.is_floating_point else F32)]
