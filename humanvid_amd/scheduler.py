"""DDIM scheduler of the CamAnimate pipeline and the context-window scheduler.

 * `DDIMScheduler`: the subset of diffusers' DDIMScheduler that configs/inference/inference_v2.yaml:
   24-33 selects (linear betas 0.00085 -> 0.012, steps_offset 1, clip_sample False, v_prediction,
   rescale_betas_zero_snr, timestep_spacing "trailing", set_alpha_to_one) -- SURVEY.md appendix C.
   Only host-side scalars live here; the tensor update itself is hv_cfg_ddim_step.
 * `uniform` / `ordered_halving` / `get_context_scheduler`: the context-window policy of
   /root/reference/src/pipelines/context.py:7-49, restated as a closed-form window table (`window_table`);
   integer-exact against the reference-generated tests/golden/context_windows.json.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, List, Optional

import numpy as np
import torch


@dataclass
class DDIMSchedulerOutput:
    prev_sample: torch.Tensor


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02,
                 beta_schedule: str = "linear", clip_sample: bool = True, set_alpha_to_one: bool = True,
                 steps_offset: int = 0, prediction_type: str = "epsilon", rescale_betas_zero_snr: bool = False,
                 timestep_spacing: str = "leading", **unused):
        if beta_schedule != "linear":
            raise NotImplementedError(beta_schedule)
        if prediction_type not in ("v_prediction", "epsilon"):
            raise NotImplementedError(prediction_type)
        if clip_sample:
            raise NotImplementedError("clip_sample=True is not used by the CamAnimate configs")
        self.num_train_timesteps = num_train_timesteps
        self.prediction_type = prediction_type
        self.timestep_spacing = timestep_spacing
        self.steps_offset = steps_offset
        betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        if rescale_betas_zero_snr:
            abar = torch.cumprod(1.0 - betas, dim=0)
            s = abar.sqrt()
            s0, sT = s[0].clone(), s[-1].clone()
            s = (s - sT) * s0 / (s0 - sT)
            abar = s**2
            alphas = torch.cat([abar[0:1], abar[1:] / abar[:-1]])
            betas = 1 - alphas
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        T = self.num_train_timesteps
        if self.timestep_spacing == "trailing":
            ts = np.round(np.arange(T, 0, -T / num_inference_steps)) - 1
        elif self.timestep_spacing == "leading":
            ts = (np.arange(0, num_inference_steps) * (T // num_inference_steps)).round()[::-1].copy() + self.steps_offset
        else:
            raise NotImplementedError(self.timestep_spacing)
        self.timesteps = torch.from_numpy(ts.astype(np.int64)).to(device) if device is not None else torch.from_numpy(
            ts.astype(np.int64))

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step_coefficients(self, timestep: int):
        """sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev) as python floats (fp32 arithmetic)."""
        t = int(timestep)
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a = self.alphas_cumprod[t]
        ap = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        return float(a.sqrt()), float((1 - a).sqrt()), float(ap.sqrt()), float((1 - ap).sqrt())

    def step(self, model_output, timestep, sample, eta: float = 0.0, **unused) -> DDIMSchedulerOutput:
        """Tensor form (any device, plain torch elementwise math) kept for API compatibility with
        callers outside the fused pipeline; the pipeline itself uses hv_cfg_ddim_step."""
        if eta != 0.0:
            raise NotImplementedError("eta > 0 (stochastic DDIM) is not used by the CamAnimate pipeline")
        sa, s1a, sap, s1ap = self.step_coefficients(int(timestep))
        if self.prediction_type == "v_prediction":
            x0 = sa * sample - s1a * model_output
            eps = sa * model_output + s1a * sample
        else:
            eps = model_output
            x0 = (sample - s1a * eps) / sa
        return DDIMSchedulerOutput(prev_sample=sap * x0 + s1ap * eps)


def fused_step_coefficients(sched, timestep: int, num_inference_steps: int):
    """(c1, c2, c3, c4) for hv_cfg_ddim_step, which evaluates  x0 = c1 x - c2 m,  eps = c1 m + c2 x,  x' = c3 x0 + c4 eps
    on the guided model output m (eta = 0, no sample clipping):

      v_prediction (inference_v2.yaml:24-33):  (sqrt a_t, sqrt(1 - a_t), sqrt a_prev, sqrt(1 - a_prev))
      epsilon (inference_v1.yaml:18-23, diffusers' default):  x' = sqrt a_prev (x - sqrt(1 - a_t) m) / sqrt a_t
          + sqrt(1 - a_prev) m is linear in (x, m) too:  (1, 0, sqrt a_prev / sqrt a_t,
          sqrt(1 - a_prev) - sqrt a_prev sqrt(1 - a_t) / sqrt a_t)  -- the same kernel, no second code path.

    `sched` is this module's DDIMScheduler or any object with diffusers' DDIMScheduler attributes (`alphas_cumprod`,
    `final_alpha_cumprod`, and `prediction_type` / `num_train_timesteps` / `clip_sample` either directly or under `.config`),
    which is what the reference's scripts construct (scripts/pose2vid.py:157-158)."""
    def attr(name, default=None):
        if hasattr(sched, name):
            return getattr(sched, name)
        cfg = getattr(sched, "config", None)
        if cfg is not None:
            return cfg[name] if isinstance(cfg, dict) and name in cfg else getattr(cfg, name, default)
        return default

    pred = attr("prediction_type", "epsilon")
    if pred not in ("v_prediction", "epsilon"):
        raise NotImplementedError(f"prediction_type {pred!r}: the fused CFG + DDIM step knows v_prediction and epsilon")
    if attr("clip_sample", False) or attr("thresholding", False):
        raise NotImplementedError("clip_sample / thresholding are not used by the CamAnimate configs and not fused")
    abar = getattr(sched, "alphas_cumprod", None)
    if abar is None:
        raise NotImplementedError("the scheduler has no alphas_cumprod: a DDIM-style scheduler is required")
    # Only DDIM's update is fused, so the check is POSITIVE (ancestral / consistency samplers such as
    # DDPMScheduler or LCMScheduler are first order, carry alphas_cumprod and no sigma table -- a blacklist lets them
    # through and samples them with DDIM eta = 0 coefficients): the scheduler must be DDIM by name or by shape -- a
    # `final_alpha_cumprod` AND a step() that takes `eta` -- and must not carry what DDIM does not have (an order above 1,
    # a sigma table, multistep solver state).
    import inspect

    name = type(sched).__name__
    step_fn = getattr(sched, "step", None)
    try:
        takes_eta = step_fn is not None and "eta" in inspect.signature(step_fn).parameters
    except (TypeError, ValueError):
        takes_eta = False
    ddim_like = "DDIM" in name or (hasattr(sched, "final_alpha_cumprod") and takes_eta)
    if not ddim_like or not hasattr(sched, "final_alpha_cumprod") or int(attr("order", 1) or 1) != 1 or \
            hasattr(sched, "sigmas") or hasattr(sched, "model_outputs") or hasattr(sched, "ets") or \
            attr("solver_order") is not None or attr("algorithm_type") is not None:
        raise NotImplementedError(f"{name} is not a DDIM scheduler (needs final_alpha_cumprod and a DDIM step(..., eta)): the fused "
                                  "CFG + DDIM step would sample it with DDIM (eta = 0) coefficients")
    T = int(attr("num_train_timesteps", len(abar)))
    t = int(timestep)
    prev = t - T // int(num_inference_steps)
    a = abar[t].float() if isinstance(abar, torch.Tensor) else torch.tensor(float(abar[t]))
    final = sched.final_alpha_cumprod
    ap = (abar[prev] if prev >= 0 else final)
    ap = ap.float() if isinstance(ap, torch.Tensor) else torch.tensor(float(ap))
    sa, s1a, sap, s1ap = float(a.sqrt()), float((1 - a).sqrt()), float(ap.sqrt()), float((1 - ap).sqrt())
    if pred == "v_prediction":
        return sa, s1a, sap, s1ap
    if sa == 0.0:
        raise ValueError("epsilon prediction at a timestep with zero signal (alpha_bar = 0): the zero-terminal-SNR schedule "
                         "needs prediction_type='v_prediction'")
    return 1.0, 0.0, sap / sa, s1ap - sap * s1a / sa


# ---------------------------------------------------------------------------------- context windows
# Behavioural contract: /root/reference/src/pipelines/context.py:7-76 (pinned by tests/golden/context_windows.json, which
# the reference's own generator produced).  Stated here as a closed-form window table:
#
#   r(step)   = the binary digits of `step` mirrored around the binary point (1 -> .1 = 1/2, 2 -> .01 = 1/4, 3 -> .11 ...):
#               a van-der-Corput sequence, so successive denoising steps start their windows at well-spread frames
#   levels    = min(context_stride, ceil(log2(F / size)) + 1)  dilation levels, level k samples every 2^k-th frame
#   level k   : windows start at  floor(r * 2^k) + round(F * r)  and advance by  size * 2^k - overlap  until the start
#               passes  F + round(F * r)  (minus `overlap` for an open loop); frame indices wrap modulo F.
def van_der_corput(n: int) -> float:
    """Radical inverse of n in base 2 (exact in binary floating point for n < 2**53)."""
    frac, weight = 0.0, 0.5
    n = int(n)
    while n:
        if n & 1:
            frac += weight
        weight *= 0.5
        n >>= 1
    return frac


ordered_halving = van_der_corput  # the reference's name for it (context.py:7-12)


def window_table(step: int, num_frames: int, context_size: int, context_stride: int = 3, context_overlap: int = 4,
                 closed_loop: bool = True) -> List[List[int]]:
    """All context windows of one denoising step as lists of frame indices."""
    F, size = int(num_frames), int(context_size)
    if F <= size:
        return [list(range(F))]
    levels = min(int(context_stride), int(math.ceil(math.log2(F / size))) + 1)
    r = van_der_corput(step)
    shift = int(round(F * r))
    stop = F + shift - (0 if closed_loop else context_overlap)
    table = []
    for k in range(levels):
        dil = 1 << k
        hop = size * dil - context_overlap
        offsets = dil * np.arange(size)
        for start in range(int(r * dil) + shift, stop, hop):
            table.append(((start + offsets) % F).tolist())
    return table


def uniform(step: int = ..., num_steps: Optional[int] = None, num_frames: int = ..., context_size: Optional[int] = None,
            context_stride: int = 3, context_overlap: int = 4, closed_loop: bool = True):
    """Generator form with the reference's signature (`num_steps` is unused there as well)."""
    yield from window_table(step, num_frames, context_size, context_stride, context_overlap, closed_loop)


_POLICIES = {"uniform": uniform}


def get_context_scheduler(name: str) -> Callable:
    try:
        return _POLICIES[name]
    except KeyError:
        raise ValueError(f"Unknown context_overlap policy {name}") from None


def get_total_steps(scheduler, timesteps: List[int], num_steps: Optional[int] = None, num_frames: int = ...,
                    context_size: Optional[int] = None, context_stride: int = 3, context_overlap: int = 4,
                    closed_loop: bool = True):
    """Number of UNet forwards of a whole sampling run (one per window per timestep)."""
    total = 0
    for i, _ in enumerate(timesteps):
        total += sum(1 for _ in scheduler(i, num_steps, num_frames, context_size, context_stride, context_overlap))
    return total
