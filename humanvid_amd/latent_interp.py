"""Latent interpolation between denoised frames -- drop-in for Pose2VideoPipeline.interpolate_latents
(/root/reference/src/pipelines/pipeline_pose2vid_long.py:294-337) and the process-wide blend selection of
/root/reference/src/pipelines/utils.py:3-30 (`set_tensor_interpolation_method(is_slerp)`; unset -> the reference fails with
"'NoneType' object is not callable" as soon as a factor >= 2 is asked for, and so does this module, by name).

A post-step of `__call__` on a [1,4,F,h,w] fp32 latent (F <= 48, 4 x 128 x 72 values per frame): host-side torch ops on
whatever device the latents live on, all (frame pair, fraction) blends in one broadcast expression instead of the
reference's per-frame Python loop."""
from __future__ import annotations

import torch

_METHOD = None  # None | "linear" | "slerp"
DOT_THRESHOLD = 0.9995


def linear(v0: torch.Tensor, v1: torch.Tensor, t: float) -> torch.Tensor:
    return (1.0 - t) * v0 + t * v1


def slerp(v0: torch.Tensor, v1: torch.Tensor, t: float, DOT_THRESHOLD: float = DOT_THRESHOLD) -> torch.Tensor:
    """spherical blend along the angle between the two whole tensors; nearly parallel tensors blend linearly"""
    cos = torch.dot((v0 / v0.norm()).flatten(), (v1 / v1.norm()).flatten())
    if cos.abs() > DOT_THRESHOLD:
        return linear(v0, v1, t)
    omega = cos.acos()
    return (torch.sin((1.0 - t) * omega) * v0 + torch.sin(t * omega) * v1) / torch.sin(omega)


def set_tensor_interpolation_method(is_slerp: bool) -> None:
    global _METHOD
    _METHOD = "slerp" if is_slerp else "linear"


def get_tensor_interpolation_method():
    return {None: None, "linear": linear, "slerp": slerp}[_METHOD]


def interpolate_latents(latents: torch.Tensor, interpolation_factor: int, device=None, method=None) -> torch.Tensor:
    """[B,C,F,h,w] -> [B,C,(F-1)*k+1,h,w]: frame i of the input lands at i*k, the k-1 frames behind it are the blends of
    frames i and i+1 at t = j/k.  `method`: "linear" | "slerp" | None (= the process-wide selection)."""
    k = int(interpolation_factor)
    if k < 2:
        return latents
    method = _METHOD if method is None else method
    if method not in ("linear", "slerp"):
        raise TypeError("'NoneType' object is not callable: call set_tensor_interpolation_method(is_slerp) before asking for "
                        "interpolation_factor >= 2 (src/pipelines/utils.py:3-12)")
    B, C, F, h, w = latents.shape
    x = latents if device is None else latents.to(device)
    a, b = x[:, :, :-1], x[:, :, 1:]                                       # [B,C,F-1,h,w] frame pairs
    t = torch.arange(k, device=x.device, dtype=x.dtype) / k                # fractions 0, 1/k .. (k-1)/k
    t = t.view(1, 1, 1, k, 1, 1)
    a6, b6 = a.unsqueeze(3), b.unsqueeze(3)                                # [B,C,F-1,1,h,w]
    wa, wb = 1.0 - t, t                                                    # linear weights, broadcast over pairs
    if method == "slerp":
        # one angle per frame pair, over ALL elements of the two frames (batch and channels included)
        na = a.pow(2).sum(dim=(0, 1, 3, 4)).sqrt()
        nb = b.pow(2).sum(dim=(0, 1, 3, 4)).sqrt()
        cos = ((a / na.view(1, 1, -1, 1, 1)) * (b / nb.view(1, 1, -1, 1, 1))).sum(dim=(0, 1, 3, 4))   # [F-1]
        flat = cos.abs() > DOT_THRESHOLD
        omega = torch.where(flat, torch.ones_like(cos), cos.clamp(-1.0, 1.0).acos()).view(1, 1, -1, 1, 1, 1)
        sa, sb = torch.sin((1.0 - t) * omega) / torch.sin(omega), torch.sin(t * omega) / torch.sin(omega)
        fl = flat.view(1, 1, -1, 1, 1, 1)
        wa, wb = torch.where(fl, wa.expand_as(sa), sa), torch.where(fl, wb.expand_as(sb), sb)
    body = (wa * a6 + wb * b6).reshape(B, C, (F - 1) * k, h, w)           # t = 0 reproduces frame i itself
    out = torch.cat([body, x[:, :, -1:]], dim=2)
    return out.to(latents.device)
