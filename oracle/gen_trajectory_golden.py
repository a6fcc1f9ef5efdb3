#!/usr/bin/env python3
"""30-step DDIM trajectory of the fp32 oracle loop at the SD-1.5 widths of the benchmarked model  --  TEST INFRASTRUCTURE ONLY.

    python oracle/gen_trajectory_golden.py        ->  tests/golden/trajectory_sd15.npz

oracle_torch.denoise_loop (the restatement of pipeline_pose2vid_long.py:454-571, pinned by oracle/gen_golden.py against the
reference's own modules) runs the reference's default sampling length -- 30 steps, CFG 3.5, v-prediction -- on the seeded inputs
of tests/test_gpu_pipeline.py::test_thirty_step_trajectory_stays_bounded[sd15] (4 frames of a 32 x 16 latent, four levels
320 / 640 / 1280 / 1280, all banked transformers and motion modules).  The latents after every step are committed as fp16 so that
the GPU test compares the native trajectory without spending four minutes of host time on the oracle in every run.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle_torch as O  # noqa: E402


def inputs():
    """the seeded inputs of the sd15 trajectory case (shared with the test through this function)"""
    cfg = dict(O.SD15_UNET3D_CFG)
    g = torch.Generator().manual_seed(77)
    F, H, W, h, w = 4, 256, 128, 32, 16
    lat = torch.randn(1, 4, F, h, w, generator=g)
    pose = torch.rand(1, 3, F, H, W, generator=g)
    pl = torch.randn(1, 6, F, H, W, generator=g)
    clip = torch.randn(1, 1, 768, generator=g)
    chan = {"320": 0, "640": 1, "1280": 2}
    banks = {}
    sd = O.make_unet3d_weights(cfg, seed=0)
    for p in O.transformer_locations(cfg):
        c = sd[p + ".norm.weight"].numel()
        lvl = 3 if p.startswith("mid_block") else chan[str(c)]
        banks[p] = torch.randn(2, (h >> lvl) * (w >> lvl), c, generator=g).half().float()
    return cfg, sd, lat, pose, pl, clip, banks


def main():
    torch.set_grad_enabled(False)
    cfg, sd, lat, pose, pl, clip, banks = inputs()
    trace = []
    t0 = time.time()
    O.denoise_loop(sd, cfg, O.make_pose_guider_weights(), O.make_camera_encoder_weights(), lat.clone(), pose, pl, clip, banks, 30, 3.5,
                   trace=trace)
    print(f"30 oracle steps in {time.time() - t0:.0f} s; final latent rms {float(trace[-1].pow(2).mean().sqrt()):.4f}")
    out = os.path.join(os.path.dirname(HERE), "tests", "golden", "trajectory_sd15.npz")
    np.savez_compressed(out, trace=torch.stack(trace).half().numpy())
    print("wrote", out)


if __name__ == "__main__":
    main()
