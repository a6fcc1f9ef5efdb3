"""Seeded inputs of the full-size parity cases (BASELINE.json configs #2, #3 and a window of #5)  --  test infrastructure.

Shared by oracle/gen_fullsize_golden.py (build container: runs the REFERENCE's own UNet3DConditionModel on these
inputs and commits the result under tests/golden/) and tests/test_gpu_fullsize_parity.py (GPU box: runs the native
path on the same inputs).  Everything is derived from CPU torch generators with fixed seeds so both sides build
bit-identical tensors; weights come from oracle_torch.make_unet3d_weights(SD15_UNET3D_CFG, seed=WEIGHT_SEED).
"""
import torch

WEIGHT_SEED = 0
TIMESTEP = 499
CASES = {
    # name: frames, latent h, latent w   (config #3: 24f x 768x512; config #2: 16f x 512x512)
    "config3": dict(F=24, h=96, w=64),
    "config2": dict(F=16, h=64, w=64),
    # one 24-frame context window of config #5 (48f x 1024x576: latent 128x72, three such windows per step)
    "config5": dict(F=24, h=128, w=72),
}


def level_of(loc: str, channels: int) -> int:
    return 3 if loc.startswith("mid_block") else {320: 0, 640: 1, 1280: 2}[channels]


def make_inputs(case: str, locations, channels_of):
    """-> sample [2,4,F,h,w], ehs [2,1,768], pose_cond_fea [2,320,F,h,w], banks {loc: [2,N_l,C] (fp16-rounded)}.
    `channels_of(loc)` gives the hidden size of a transformer location."""
    c = CASES[case]
    F, h, w = c["F"], c["h"], c["w"]
    g = torch.Generator().manual_seed(42)
    sample = torch.randn(1, 4, F, h, w, generator=g).repeat(2, 1, 1, 1, 1)
    ehs = torch.cat([torch.zeros(1, 1, 768), torch.randn(1, 1, 768, generator=torch.Generator().manual_seed(2))])
    pose = (torch.randn(1, 320, F, h, w, generator=torch.Generator().manual_seed(1)) * 0.5).repeat(2, 1, 1, 1, 1)
    gb = torch.Generator().manual_seed(5)
    banks = {}
    for loc in locations:
        C = channels_of(loc)
        lvl = level_of(loc, C)
        banks[loc] = torch.randn(2, (h >> lvl) * (w >> lvl), C, generator=gb).half().float()
    return sample, ehs, pose, banks


def tap_images(F: int):
    """(batch, frame) pairs whose activations are kept: one unconditional, one conditional image."""
    return [(0, 1), (1, F - 2)]


def tap_grid(hl: int, wl: int):
    return list(range(0, hl, max(1, hl // 12))), list(range(0, wl, max(1, wl // 8)))


def slice_ncfhw(x: torch.Tensor, F: int) -> torch.Tensor:
    """x [b,c,f,h,w] -> [2, ny, nx, c] (NHWC slices of the two tap images)."""
    ys, xs = tap_grid(x.shape[3], x.shape[4])
    out = []
    for (bi, fi) in tap_images(F):
        img = x[bi, :, fi]  # [c,h,w]
        out.append(img[:, ys][:, :, xs].permute(1, 2, 0))
    return torch.stack(out)


def slice_nhwc(x: torch.Tensor, F: int) -> torch.Tensor:
    """x [(b f),h,w,c] -> [2, ny, nx, c]."""
    ys, xs = tap_grid(x.shape[1], x.shape[2])
    out = []
    for (bi, fi) in tap_images(F):
        img = x[bi * F + fi]
        out.append(img[ys][:, xs])
    return torch.stack(out)


def rms_ncfhw(x: torch.Tensor) -> torch.Tensor:
    """per-image rms, [b*f] in (b f) order."""
    return x.float().pow(2).mean(dim=(1, 3, 4)).sqrt().reshape(-1)


def rms_nhwc(x: torch.Tensor) -> torch.Tensor:
    return x.float().pow(2).mean(dim=(1, 2, 3)).sqrt()


# ---- loop-body cases (oracle/gen_fullsize_steps_golden.py <-> tests/test_gpu_fullsize_steps.py) ----------------------
# "steps3":   two consecutive denoising steps (i = 0: t = 999, zero terminal SNR, sqrt(abar) = 0; i = 1: t = 966) of the
#             30-step schedule at config #3's size, one 24-frame window, PoseGuider + CameraPoseEncoder in the loop
# "windows48": one step of a 48-frame clip (context 24, overlap 4 -> three overlapping windows, the geometry of config #5's
#             step) at a reduced 32 x 32 latent: window accumulation + counter division + CFG + DDIM
# "edge_t32": the LAST step of the 30-step schedule (t = 32 -> prev < 0 -> final_alpha_cumprod) at config #3's size
# "traj30":   all 30 steps at config #3's size from the same seeded start: the trajectory the benchmark times, end to end
STEP_CASES = {
    "steps3": dict(F=24, h=96, w=64, steps=(0, 1), num_inference_steps=30),
    "windows48": dict(F=48, h=32, w=32, steps=(0, 1), num_inference_steps=30),
    "edge_t32": dict(F=24, h=96, w=64, steps=(29,), num_inference_steps=30),
    # the WHOLE 30-step trajectory at config #3's size (2.5 h of host time once); only the latents after the steps in `keep`
    # are committed (fp16)
    "traj30": dict(F=24, h=96, w=64, steps=tuple(range(30)), num_inference_steps=30, keep=(0, 4, 9, 14, 19, 24, 29)),
}
GUIDANCE = 3.5


def make_step_inputs(case: str, locations, channels_of):
    """-> latents [1,4,F,h,w], pose images [1,3,F,8h,8w] in [0,1], Pluecker map [1,6,F,8h,8w], clip [1,1,768],
    banks {loc: [2,N_l,C]} -- all from CPU generators with fixed seeds."""
    c = STEP_CASES[case]
    F, h, w = c["F"], c["h"], c["w"]
    g = torch.Generator().manual_seed(1234)
    lat = torch.randn(1, 4, F, h, w, generator=g)
    pose = torch.rand(1, 3, F, 8 * h, 8 * w, generator=torch.Generator().manual_seed(11))
    pl = torch.randn(1, 6, F, 8 * h, 8 * w, generator=torch.Generator().manual_seed(12))
    clip = torch.randn(1, 1, 768, generator=torch.Generator().manual_seed(13))
    gb = torch.Generator().manual_seed(15)
    banks = {}
    for loc in locations:
        C = channels_of(loc)
        lvl = level_of(loc, C)
        banks[loc] = torch.randn(2, (h >> lvl) * (w >> lvl), C, generator=gb).half().float()
    return lat, pose, pl, clip, banks
