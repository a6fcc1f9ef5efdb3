"""Algorithmic work of one UNet3D forward (FLOP / byte bookkeeping for roofline reporting).

"minimal" = useful work only: the CFG-unconditional half attends N keys, the
conditional half 2N (own + bank); bank K/V projected once per clip; the one-key CLIP cross-attention
folded to a constant.  "as_written" = what the reference's PyTorch code executes per step
(mutual_self_attention.py:147-186 computes all frames against 2N keys and recomputes the
unconditional half; attention.py:410-424 runs the 1-key cross-attention in full).
"""
from __future__ import annotations

from typing import Dict

from .arch import build_block_specs


def unet3d_flops(cfg: dict, B: int, F: int, h: int, w: int, as_written: bool = False) -> Dict[str, float]:
    n = B * F
    out = dict(conv=0.0, attn=0.0, attn_linear=0.0, ff=0.0, motion_linear=0.0, motion_ff=0.0, temporal=0.0, cross=0.0)
    boc = tuple(cfg["block_out_channels"])
    out["conv"] += 2.0 * n * h * w * 9 * cfg["in_channels"] * boc[0]
    out["conv"] += 2.0 * n * h * w * 9 * boc[0] * cfg["out_channels"]
    hh, ww = h, w
    n_motion_attn = len(cfg.get("motion_module_kwargs", {}).get("attention_block_types", ()))

    def transformer(C, N):
        M = n * N
        out["attn_linear"] += 2.0 * M * C * C * 2  # proj_in, proj_out
        out["attn_linear"] += 2.0 * M * C * 3 * C + 2.0 * M * C * C  # qkv, out
        half = n / 2 if B == 2 else 0
        if as_written and B == 2:
            out["attn"] += 4.0 * N * 2 * N * C * n + 4.0 * N * N * C * half
            out["attn_linear"] += 2.0 * M * C * 2 * C  # k, v of the bank tokens repeated per frame
            out["attn_linear"] += 2.0 * (M / 2) * C * 4 * C  # unconditional half recomputed (q,k,v,out)
            out["cross"] += 2.0 * M * C * C * 2 + 4.0 * N * C * n
        else:
            out["attn"] += 4.0 * N * N * C * half + 4.0 * N * 2 * N * C * (n - half)
        out["ff"] += 2.0 * M * C * 8 * C + 2.0 * M * 4 * C * C

    def motion(C, N):
        M = n * N
        out["motion_linear"] += 2.0 * M * C * C * 2
        out["motion_linear"] += n_motion_attn * (2.0 * M * C * 3 * C + 2.0 * M * C * C)
        out["temporal"] += n_motion_attn * 4.0 * F * F * C * B * N
        out["motion_ff"] += 2.0 * M * C * 8 * C + 2.0 * M * 4 * C * C

    for spec in build_block_specs(cfg):
        C = spec.channels
        n_attn = len(spec.resnets) - 1 if spec.kind == "mid" else len(spec.resnets)
        for j, (m, s, o) in enumerate(spec.resnets):
            M = n * hh * ww
            out["conv"] += 2.0 * M * 9 * (m + s) * o + 2.0 * M * 9 * o * o
            if m + s != o:
                out["conv"] += 2.0 * M * (m + s) * o
            if j < n_attn:
                if spec.has_attn:
                    transformer(C, hh * ww)
                if spec.has_motion:
                    motion(C, hh * ww)
        if spec.kind == "down" and spec.resample:
            hh, ww = (hh + 1) // 2, (ww + 1) // 2
            out["conv"] += 2.0 * n * hh * ww * 9 * C * C
        if spec.kind == "up" and spec.resample:
            hh, ww = hh * 2, ww * 2
            out["conv"] += 2.0 * n * hh * ww * 9 * C * C
    out["total"] = sum(out.values())
    return out
