"""Parity of the conditioning encoders and of the whole denoising loop on MI355X.

 * PoseGuider / CameraPoseEncoder against golden fixtures produced by the REFERENCE modules
   (tests/golden/pose_guider.npz, camera_encoder.npz; oracle/gen_golden.py),
 * Pose2VideoPipeline.denoise (hoisted conditioning, HIP-graph replay, context windows, CFG + DDIM)
   against the oracle's restatement of pipeline_pose2vid_long.py:454-571, step by step.

Stated tolerance: bf16 storage vs the fp32 reference -> NRMSE <= 2e-2 per UNet evaluation; the
latents after k DDIM steps accumulate at most that relative deviation per step (checked <= 2e-2).
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import oracle_torch as O  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def nrmse(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm())


def make_pose_guider():
    from humanvid_amd.conditioning import PoseGuider

    pg = PoseGuider(320, block_out_channels=(16, 32, 96, 256))
    sd = O.make_pose_guider_weights()
    pg.load_state_dict(sd, strict=True)
    return pg.to("cuda"), sd


def make_camera_encoder():
    from humanvid_amd.conditioning import CameraPoseEncoder

    cam = CameraPoseEncoder(downscale_factor=8, channels=[320], nums_rb=2, cin=384, ksize=1, sk=True, use_conv=False,
                            compression_factor=1, temporal_attention_nhead=8, attention_block_types=["Temporal_Self"],
                            temporal_position_encoding=True, temporal_position_encoding_max_len=24)
    sd = O.make_camera_encoder_weights()
    cam.load_state_dict(sd, strict=True)
    return cam.to("cuda"), sd


def test_pose_guider_golden():
    z = np.load(os.path.join(GOLD, "pose_guider.npz"))
    pg, _ = make_pose_guider()
    out = pg(torch.from_numpy(z["cond"]).cuda())
    e = nrmse(out, torch.from_numpy(z["out"]))
    print("pose guider vs reference golden nrmse", e)
    assert e < 2e-2


def test_camera_encoder_golden():
    z = np.load(os.path.join(GOLD, "camera_encoder.npz"))
    cam, _ = make_camera_encoder()
    out = cam(torch.from_numpy(z["plucker"]).cuda())[0]
    e = nrmse(out, torch.from_numpy(z["out"]))
    print("camera encoder vs reference golden nrmse", e)
    assert e < 2e-2


def test_camera_front_end_on_device_matches_plucker_map_path():
    """CameraPoseEncoder fed (K, c2w) -- Pluecker map generated inside the PixelUnshuffle
    kernel -- must reproduce the path that is fed the materialised ray_condition map (oracle restatement of
    dance_image_h_v_camera.py:88-130, pinned against the reference by oracle/gen_golden.py)."""
    cam, sd = make_camera_encoder()
    g = torch.Generator().manual_seed(3)
    Fc, H, W = 4, 64, 48
    K = torch.tensor([[60.0, 58.0, 24.0, 32.0]]).repeat(Fc, 1) + torch.rand(Fc, 4, generator=g)
    c2w = torch.eye(4).repeat(Fc, 1, 1)
    for i in range(1, Fc):
        c2w[i, :3, :3] = torch.linalg.qr(torch.eye(3) + 0.1 * torch.randn(3, 3, generator=g))[0]
        c2w[i, :3, 3] = 0.3 * torch.randn(3, generator=g)
    pl = O.ray_condition(K[None], c2w[None], H, W)[0].permute(3, 0, 1, 2)[None]  # [1,6,F,H,W]
    add = (torch.randn(Fc, H // 8, W // 8, 320, generator=g) * 0.1).to(torch.bfloat16).cuda()
    a = cam.forward_nhwc(pl.cuda(), add=add).float().cpu()
    b = cam.forward_nhwc_from_cameras(K, c2w, H, W, add=add).float().cpu()
    ref = O.camera_encoder_forward(sd, pl)  # [F,320,h,w] fp32 oracle (without the added pose feature)
    ref = ref.permute(0, 2, 3, 1) + add.float().cpu()
    e_ab, e_ref = nrmse(b, a), nrmse(b, ref)
    print("camera front-end on device vs map path nrmse", e_ab, "vs oracle", e_ref)
    assert e_ab < 5e-3 and e_ref < 2e-2


@pytest.mark.parametrize("case", ["single_window_graph", "single_window_eager", "windows"])
def test_denoise_loop_matches_oracle(case):
    from humanvid_amd.pipeline import Pose2VideoPipeline
    from humanvid_amd.scheduler import DDIMScheduler
    from humanvid_amd.unet3d import UNet3DConditionModel

    cfg = O.tiny_unet3d_cfg()
    sd = O.make_unet3d_weights(cfg, seed=0)
    net = UNet3DConditionModel(**dict(cfg, use_inflated_groupnorm=True, unet_use_cross_frame_attention=False,
                                      unet_use_temporal_attention=False, motion_module_type="Vanilla"))
    net.load_state_dict(sd, strict=True)
    net = net.to("cuda")
    pg, pg_sd = make_pose_guider()
    cam, cam_sd = make_camera_encoder()
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=False, steps_offset=1,
                          prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")
    pipe = Pose2VideoPipeline(None, None, None, net, pg, cam, sched)
    g = torch.Generator().manual_seed(42)
    if case == "windows":
        F, cf, ov = 12, 8, 2
    else:
        F, cf, ov = 4, 24, 4
    H = W = 64
    h = w = 8
    lat = torch.randn(1, 4, F, h, w, generator=g)
    pose = torch.rand(1, 3, F, H, W, generator=g)
    pl = torch.randn(1, 6, F, H, W, generator=g)
    clip = torch.randn(1, 1, 768, generator=g)
    banks = {}
    for p in O.transformer_locations(cfg):
        c = sd[p + ".norm.weight"].numel()
        banks[p] = torch.randn(2, h * w if c == 320 else (h // 2) * (w // 2), c, generator=g).half().float()
    steps, run_steps = 4, 3
    trace = []
    O.denoise_loop(sd, cfg, pg_sd, cam_sd, lat.clone(), pose, pl, clip, banks, steps, 3.5, context_frames=cf,
                   context_stride=1, context_overlap=ov, max_steps=run_steps, trace=trace)
    eng = net.engine()
    eng.set_reference_banks({k: v.cuda() for k, v in banks.items()}, do_cfg=True)
    eng._banks_from_modules = lambda: None
    got = []
    out = pipe.denoise(lat.clone().cuda(), pose.cuda(), pl.cuda(), clip.cuda(), steps, 3.5, context_frames=cf,
                       context_stride=1, context_overlap=ov, use_graph=(case != "single_window_eager"),
                       max_steps=run_steps, callback=lambda i, t, x: got.append(x.detach().float().cpu().clone()))
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    errs = [nrmse(a, b) for a, b in zip(got, trace)]
    print(case, "latent nrmse per step", errs)
    assert len(errs) == run_steps and max(errs) < 2e-2, errs


@pytest.mark.parametrize("widths", ["tiny", "sd15"])
def test_thirty_step_trajectory_stays_bounded(widths):
    """A full 30-step DDIM trajectory (the reference's default sampling length, scripts/pose2vid.py): the native latents are
    compared with the fp32 oracle loop after EVERY step.  Per-evaluation error of the bf16 path is ~1e-2 (tests above); DDIM
    with v-prediction re-injects only sqrt(1-a_prev)*sa - ... of it per step, so the deviation must not compound.  Stated
    bound: latent NRMSE <= 4e-2 at every one of the 30 steps and <= 4e-2 at the end (measured values are printed).
    "tiny": the two-level model of the other tests on an 8 x 8 latent, oracle loop run here.  "sd15": the SD-1.5 widths of
    the benchmarked model (four levels 320 / 640 / 1280 / 1280, all 16 banked transformers and 21 motion modules), 4 frames
    of a 32 x 16 latent, against the committed oracle trajectory tests/golden/trajectory_sd15.npz (oracle/
    gen_trajectory_golden.py: the same oracle loop, run once in the build container -- four minutes of host time per run
    otherwise)."""
    from humanvid_amd.pipeline import Pose2VideoPipeline
    from humanvid_amd.scheduler import DDIMScheduler
    from humanvid_amd.unet3d import UNet3DConditionModel

    pg, pg_sd = make_pose_guider()
    cam, cam_sd = make_camera_encoder()
    if widths == "tiny":
        cfg = O.tiny_unet3d_cfg()
        sd = O.make_unet3d_weights(cfg, seed=0)
        g = torch.Generator().manual_seed(77)
        F, H, W, h, w = 4, 64, 64, 8, 8
        lat = torch.randn(1, 4, F, h, w, generator=g)
        pose = torch.rand(1, 3, F, H, W, generator=g)
        pl = torch.randn(1, 6, F, H, W, generator=g)
        clip = torch.randn(1, 1, 768, generator=g)
        banks = {}
        for p in O.transformer_locations(cfg):
            c = sd[p + ".norm.weight"].numel()
            banks[p] = torch.randn(2, h * w if c == 320 else (h // 2) * (w // 2), c, generator=g).half().float()
        trace = []
        O.denoise_loop(sd, cfg, pg_sd, cam_sd, lat.clone(), pose, pl, clip, banks, 30, 3.5, trace=trace)
    else:
        import gen_trajectory_golden as GT  # oracle/: the seeded inputs of the committed trajectory

        cfg, sd, lat, pose, pl, clip, banks = GT.inputs()
        z = np.load(os.path.join(os.path.dirname(__file__), "golden", "trajectory_sd15.npz"))
        trace = [torch.from_numpy(t.astype(np.float32)) for t in z["trace"]]
    net = UNet3DConditionModel(**dict(cfg, use_inflated_groupnorm=True, unet_use_cross_frame_attention=False,
                                      unet_use_temporal_attention=False, motion_module_type="Vanilla"))
    net.load_state_dict(sd, strict=True)
    net = net.to("cuda")
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=False, steps_offset=1,
                          prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")
    pipe = Pose2VideoPipeline(None, None, None, net, pg, cam, sched)
    eng = net.engine()
    eng.set_reference_banks({k: v.cuda() for k, v in banks.items()}, do_cfg=True)
    eng._banks_from_modules = lambda: None
    got = []
    pipe.denoise(lat.clone().cuda(), pose.cuda(), pl.cuda(), clip.cuda(), 30, 3.5,
                 callback=lambda i, t, x: got.append(x.detach().float().cpu().clone()))
    torch.cuda.synchronize()
    errs = [nrmse(a, b) for a, b in zip(got, trace)]
    print(f"30-step trajectory ({widths}), latent nrmse per step:", " ".join(f"{e:.4f}" for e in errs))
    assert len(errs) == 30 and all(torch.isfinite(x).all() for x in got)
    assert max(errs) < 4e-2 and errs[-1] < 4e-2, errs
