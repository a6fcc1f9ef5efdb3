"""End-to-end `__call__` of the drop-in pipelines on MI355X, PIL images in -> videos out, against the oracle.

The reference's entry points (scripts/pose2vid.py:176-295 -> pipeline_pose2vid_long.py:339-588; scripts/pose2img.py ->
pipeline_pose2img.py:195-376; pipeline_pose2vid.py:285-458) are executed here through the native package exactly as a
script would call them: CLIP embedding -> ReferenceNet write pass -> bank hand-over -> denoising loop -> VAE decode.
CLIP and VAE stay PyTorch modules supplied by the caller (BASELINE.json north_star); small deterministic stand-ins with
the same interfaces are used (no checkpoints are reachable).  The oracle side restates the same call on the CPU in fp32:
same stand-ins, oracle ReferenceNet banks (rounded through fp16 as update() does), oracle denoising loop, same decode.

Stated tolerances: latents after every DDIM step NRMSE <= 2e-2 (bf16 path vs fp32 oracle), decoded video NRMSE <= 3e-2,
values in [0, 1], shapes as the reference returns them.
"""
import os
import sys

import numpy as np
import pytest
import torch
from torch import nn

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import oracle_torch as O  # noqa: E402

pytestmark = pytest.mark.gpu


class _Dist:
    def __init__(self, mean):
        self.mean = mean


class _Enc:
    def __init__(self, mean):
        self.latent_dist = _Dist(mean)


class _Dec:
    def __init__(self, sample):
        self.sample = sample


class _Cfg:
    block_out_channels = (1, 1, 1, 1)  # -> vae_scale_factor 8


class TinyVAE(nn.Module):
    """AutoencoderKL stand-in: 8x down / up, frames independent."""

    config = _Cfg()

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(11)
        self.enc = nn.Conv2d(3, 4, 8, stride=8)
        self.dec = nn.ConvTranspose2d(4, 3, 8, stride=8)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.08)

    @property
    def dtype(self):
        return self.enc.weight.dtype

    @property
    def device(self):
        return self.enc.weight.device

    def encode(self, x):
        return _Enc(self.enc(x))

    def decode(self, z):
        return _Dec(self.dec(z))


class _Emb:
    def __init__(self, e):
        self.image_embeds = e


class TinyCLIP(nn.Module):
    """CLIPVisionModelWithProjection stand-in: [1,3,224,224] -> image_embeds [1,768]."""

    def __init__(self):
        super().__init__()
        self.proj = nn.Linear(3 * 7 * 7, 768)
        with torch.no_grad():
            g = torch.Generator().manual_seed(12)
            self.proj.weight.copy_(torch.randn(self.proj.weight.shape, generator=g) * 0.2)
            self.proj.bias.zero_()

    @property
    def dtype(self):
        return self.proj.weight.dtype

    def forward(self, x):
        return _Emb(self.proj(torch.nn.functional.adaptive_avg_pool2d(x, 7).flatten(1)))


def _pil(seed, w, h):
    from PIL import Image

    rng = np.random.default_rng(seed)
    return Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8))


def nrmse(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return float((a - b).norm() / b.norm())


def _scheduler():
    from humanvid_amd.scheduler import DDIMScheduler

    return DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=False, steps_offset=1,
                         prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")


def _build(cfg, seed, motion=True):
    """native denoising UNet + ReferenceNet with oracle weights -> (net, ref_net, sd, ref_sd)"""
    from humanvid_amd.unet2d import UNet2DConditionModel
    from humanvid_amd.unet3d import UNet3DConditionModel

    sd = O.make_unet3d_weights(cfg, seed=seed)
    kw = dict(cfg, use_inflated_groupnorm=True, unet_use_cross_frame_attention=False, unet_use_temporal_attention=False,
              motion_module_type="Vanilla" if motion else None)
    if not motion:
        kw["motion_module_kwargs"] = {}
    net = UNet3DConditionModel(**kw)
    assert not any(net.load_state_dict(sd, strict=True))
    ref_sd = O.make_reference_net_weights(cfg, seed=seed + 1)
    d2 = tuple(t.replace("3D", "2D") for t in cfg["down_block_types"])
    u2 = tuple(t.replace("3D", "2D") for t in cfg["up_block_types"])
    ref = UNet2DConditionModel(block_out_channels=cfg["block_out_channels"], layers_per_block=cfg["layers_per_block"],
                               cross_attention_dim=768, attention_head_dim=8, down_block_types=d2, up_block_types=u2)
    assert not any(ref.load_state_dict(ref_sd, strict=True))
    return net.to("cuda"), ref.to("cuda"), sd, ref_sd


def _oracle_front_end(vae, clip, ref_image, width, height, cfg, ref_sd):
    from humanvid_amd.pipeline import _clip_preprocess, _pil_to_tensor

    with torch.no_grad():
        e = clip(_clip_preprocess(ref_image)).image_embeds[:, None]  # [1,1,768]
        ref_lat = vae.encode(_pil_to_tensor(ref_image, height, width, normalize=True)).latent_dist.mean * 0.18215
        ehs = torch.cat([torch.zeros_like(e), e])
        banks = O.reference_net_banks(ref_sd, cfg, ref_lat.repeat(2, 1, 1, 1), ehs)
    return e, {k: v.half().float() for k, v in banks.items()}


def _decode(vae, latents):
    with torch.no_grad():
        f = latents.shape[2]
        flat = (latents / 0.18215).permute(0, 2, 1, 3, 4).reshape(f, 4, *latents.shape[3:])
        v = vae.decode(flat).sample
        return (v.view(1, f, *v.shape[1:]).permute(0, 2, 1, 3, 4) / 2 + 0.5).clamp(0, 1)


def test_pose2video_long_call_end_to_end():
    """Pose2VideoPipeline.__call__ (pipeline_pose2vid_long.py:339-588), two context windows per step, twice in a row with
    different reference images on one pipeline object (the scripts loop over test cases: ADVICE round 1, stale caches)."""
    from humanvid_amd.conditioning import CameraPoseEncoder, PoseGuider
    from humanvid_amd.pipeline import Pose2VideoPipeline, _pil_to_tensor

    cfg = O.tiny_unet3d_cfg()
    net, ref, sd, ref_sd = _build(cfg, seed=21)
    pg = PoseGuider(320, block_out_channels=(16, 32, 96, 256))
    pg_sd = O.make_pose_guider_weights()
    pg.load_state_dict(pg_sd, strict=True)
    cam = CameraPoseEncoder(downscale_factor=8, channels=[320], nums_rb=2, cin=384, ksize=1, sk=True, use_conv=False,
                            compression_factor=1, temporal_attention_nhead=8, attention_block_types=["Temporal_Self"],
                            temporal_position_encoding=True, temporal_position_encoding_max_len=24)
    cam_sd = O.make_camera_encoder_weights()
    cam.load_state_dict(cam_sd, strict=True)
    vae_cpu, clip_cpu = TinyVAE(), TinyCLIP()
    vae, clip = TinyVAE().to("cuda"), TinyCLIP().to("cuda")
    pipe = Pose2VideoPipeline(vae, clip, ref, net, pg.to("cuda"), cam.to("cuda"), _scheduler())
    W = H = 64
    F, steps = 6, 3
    poses = [_pil(100 + i, W, H) for i in range(F)]
    plucker = torch.randn(1, 6, F, H, W, generator=torch.Generator().manual_seed(5))
    for trial, ref_seed in enumerate((1, 2)):
        ref_image = _pil(ref_seed, 80, 96)  # resized by the pipeline
        got = []
        out = pipe(ref_image, poses, plucker, W, H, F, steps, 3.5, generator=torch.Generator().manual_seed(7),
                   context_frames=4, context_overlap=2,
                   callback=lambda i, t, x: got.append(x.detach().float().cpu().clone()))
        videos = out.videos
        assert tuple(videos.shape) == (1, 3, F, H, W) and videos.dtype == torch.float32
        assert float(videos.min()) >= 0.0 and float(videos.max()) <= 1.0
        # ---- oracle restatement of the same call
        e, banks = _oracle_front_end(vae_cpu, clip_cpu, ref_image, W, H, cfg, ref_sd)
        lat0 = torch.randn(1, 4, F, H // 8, W // 8, generator=torch.Generator().manual_seed(7))
        pose_cond = _pil_to_tensor(poses, H, W, normalize=False).permute(1, 0, 2, 3)[None]
        trace = []
        final = O.denoise_loop(sd, cfg, pg_sd, cam_sd, lat0, pose_cond, plucker, e, banks, steps, 3.5, context_frames=4,
                               context_stride=1, context_overlap=2, trace=trace)
        errs = [nrmse(a, b) for a, b in zip(got, trace)]
        ev = nrmse(videos, _decode(vae_cpu, final))
        print(f"call #{trial}: latent nrmse per step {errs}, video nrmse {ev:.3e}")
        assert len(errs) == steps and max(errs) < 2e-2, errs
        assert ev < 3e-2, ev
        assert all(len(m.bank) == 0 for m in net.modules() if hasattr(m, "bank"))  # reader.clear()


def test_pose2image_call_config1():
    """BASELINE.json configs[0]: scripts/pose2img.py geometry -- one frame, 256x256, 4 DDIM steps, CFG 3.5, SD-1.5 widths,
    no motion module -- through Pose2ImagePipeline.__call__ (pipeline_pose2img.py:195-376)."""
    from humanvid_amd.conditioning import CameraPoseEncoder, PoseGuider
    from humanvid_amd.pipeline import Pose2ImagePipeline, _pil_to_tensor

    cfg = dict(O.SD15_UNET3D_CFG, use_motion_module=False)
    net, ref, sd, ref_sd = _build(cfg, seed=31, motion=False)
    pg = PoseGuider(320, block_out_channels=(16, 32, 96, 256))
    pg_sd = O.make_pose_guider_weights()
    pg.load_state_dict(pg_sd, strict=True)
    cam = CameraPoseEncoder(downscale_factor=8, channels=[320], nums_rb=2, cin=384, ksize=1, sk=True, use_conv=False,
                            compression_factor=1, temporal_attention_nhead=8, attention_block_types=["Temporal_Self"],
                            temporal_position_encoding=True, temporal_position_encoding_max_len=24)
    cam_sd = O.make_camera_encoder_weights()
    cam.load_state_dict(cam_sd, strict=True)
    vae_cpu, clip_cpu = TinyVAE(), TinyCLIP()
    pipe = Pose2ImagePipeline(TinyVAE().to("cuda"), TinyCLIP().to("cuda"), ref, net, pg.to("cuda"), cam.to("cuda"),
                              _scheduler())
    W = H = 256
    steps = 4
    ref_image, pose_image = _pil(3, W, H), _pil(4, W, H)
    plucker = torch.randn(1, 6, H, W, generator=torch.Generator().manual_seed(6))
    got = []
    out = pipe(ref_image, pose_image, plucker, W, H, steps, 3.5, generator=torch.Generator().manual_seed(9),
               callback=lambda i, t, x: got.append(x.detach().float().cpu().clone()))
    images = out.images
    assert tuple(images.shape) == (1, 3, 1, H, W)
    assert float(images.min()) >= 0.0 and float(images.max()) <= 1.0
    e, banks = _oracle_front_end(vae_cpu, clip_cpu, ref_image, W, H, cfg, ref_sd)
    lat0 = torch.randn(1, 4, H // 8, W // 8, generator=torch.Generator().manual_seed(9)).unsqueeze(2)
    pose_cond = _pil_to_tensor(pose_image, H, W, normalize=False).unsqueeze(2)
    trace = []
    final = O.denoise_loop(sd, cfg, pg_sd, cam_sd, lat0, pose_cond, plucker.unsqueeze(2), e, banks, steps, 3.5,
                           context_frames=1, context_stride=1, context_overlap=0, trace=trace)
    errs = [nrmse(a, b) for a, b in zip(got, trace)]
    ev = nrmse(images, _decode(vae_cpu, final))
    print(f"pose2img 256x256, 4 steps: latent nrmse per step {errs}, image nrmse {ev:.3e}")
    assert len(errs) == steps and max(errs) < 2e-2, errs
    assert ev < 3e-2, ev


def test_pose2video_short_call_no_camera():
    """pipeline_pose2vid.py:285-458: all frames in one forward, pose feature only (BASELINE.json configs[1] mode)."""
    from humanvid_amd.conditioning import PoseGuider
    from humanvid_amd.pipeline import Pose2VideoShortPipeline, _pil_to_tensor

    cfg = O.tiny_unet3d_cfg()
    net, ref, sd, ref_sd = _build(cfg, seed=41)
    pg = PoseGuider(320, block_out_channels=(16, 32, 96, 256))
    pg_sd = O.make_pose_guider_weights()
    pg.load_state_dict(pg_sd, strict=True)
    vae_cpu, clip_cpu = TinyVAE(), TinyCLIP()
    pipe = Pose2VideoShortPipeline(TinyVAE().to("cuda"), TinyCLIP().to("cuda"), ref, net, pg.to("cuda"), _scheduler())
    W, H, F, steps = 64, 64, 5, 4
    ref_image = _pil(8, W, H)
    poses = [_pil(200 + i, W, H) for i in range(F)]
    got = []
    out = pipe(ref_image, poses, W, H, F, steps, 3.5, generator=torch.Generator().manual_seed(3),
               callback=lambda i, t, x: got.append(x.detach().float().cpu().clone()))
    assert tuple(out.videos.shape) == (1, 3, F, H, W)
    e, banks = _oracle_front_end(vae_cpu, clip_cpu, ref_image, W, H, cfg, ref_sd)
    lat0 = torch.randn(1, 4, F, H // 8, W // 8, generator=torch.Generator().manual_seed(3))
    pose_cond = _pil_to_tensor(poses, H, W, normalize=False).permute(1, 0, 2, 3)[None]
    # the oracle loop adds the camera feature: a zero Pluecker map through a zero-initialised zero-conv contributes nothing
    cam_sd = O.make_camera_encoder_weights()
    cam_sd = {k: (torch.zeros_like(v) if k.startswith("zero_conv_layers") else v) for k, v in cam_sd.items()}
    trace = []
    final = O.denoise_loop(sd, cfg, pg_sd, cam_sd, lat0, pose_cond, torch.zeros(1, 6, F, H, W), e, banks, steps, 3.5,
                           context_frames=F, context_stride=1, context_overlap=0, trace=trace)
    errs = [nrmse(a, b) for a, b in zip(got, trace)]
    ev = nrmse(out.videos, _decode(vae_cpu, final))
    print(f"short pipeline: latent nrmse per step {errs}, video nrmse {ev:.3e}")
    assert max(errs) < 2e-2 and ev < 3e-2, (errs, ev)
