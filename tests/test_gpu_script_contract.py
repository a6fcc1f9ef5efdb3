"""The reference's own inference script, UNMODIFIED, on the native package ("scripts/pose2vid.py runs
unchanged").

`/root/reference/scripts/pose2vid.py` is executed with `runpy` exactly as `python scripts/pose2vid.py --config ...` would be:
its `src.*` / `configs.*` imports resolve to this repository's drop-in modules, and it drives them through its whole flow --
`OmegaConf.load`, `AutoencoderKL.from_pretrained`, `UNet2DConditionModel.from_pretrained(subfolder="unet")`,
`UNet3DConditionModel.from_pretrained_2d(...)`, `PoseGuider(...)`, `CameraPoseEncoder(**pose_encoder_kwargs)`, the four
`load_state_dict` calls, `Pose2VideoPipeline(...)`, `pipe.to("cuda", dtype=fp16)`, `read_frames` / `get_fps`, the static-camera
Pluecker embedding (`Camera`, `ray_condition`), `pipe(ref_image, pose_list, camera_embedding, width, height, seq_len, steps, cfg,
generator=...)`, `save_videos_grid` -- on MI355X.  What this image lacks and the script imports at module level is stubbed in
`sys.modules` for the run: `diffusers` (AutoencoderKL -> a small stand-in VAE, DDIMScheduler -> humanvid_amd.scheduler),
`transformers` (CLIPVisionModelWithProjection -> a small stand-in), `omegaconf` (yaml + attribute dicts), `torchvision.transforms`
(PIL + numpy) and `av` (a container that keeps frames in a pickle).  Checkpoints are synthetic (no network): four-level UNets
of width 320 with seeded weights written in the layouts the script loads.

The script text is NOT part of this repository (reference sources are never copied).  The test takes it from
/root/reference when that tree exists (build container), otherwise from the environment variable HV_REF_SCRIPT_B64 (base64 of
the file), which the builder passes on the gpurun command line:

    gpurun -- "HV_REF_SCRIPT_B64=$(base64 -w0 /root/reference/scripts/pose2vid.py) python -m pytest tests/test_gpu_script_contract.py -s"

and is skipped when neither is available (the driver's GPU box).  The run of this round is recorded in
profiles/r03_script_contract.txt.
"""
import base64
import io
import json
import os
import pickle
import runpy
import sys
import types

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, os.path.dirname(__file__))

pytestmark = pytest.mark.gpu
REF_SCRIPT = "/root/reference/scripts/pose2vid.py"


def _script_text():
    if os.path.exists(REF_SCRIPT):
        return open(REF_SCRIPT).read()
    b64 = os.environ.get("HV_REF_SCRIPT_B64")
    return base64.b64decode(b64).decode() if b64 else None


# ---------------------------------------------------------------------------------------------------------- stub modules
class _AttrDict(dict):
    """omegaconf.DictConfig stand-in: attribute and item access, nested"""

    def __getattr__(self, k):
        try:
            return _wrap(self[k])
        except KeyError as e:
            raise AttributeError(k) from e

    def __getitem__(self, k):
        return _wrap(dict.__getitem__(self, k))


def _wrap(v):
    return _AttrDict(v) if isinstance(v, dict) and not isinstance(v, _AttrDict) else v


def _plain(v):
    if isinstance(v, dict):
        return {k: _plain(dict.__getitem__(v, k)) for k in v}
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    return v


class _FakeAv:
    """`av` stand-in: a "video file" is a pickle {fps, frames: [PNG bytes]}; enough of PyAV's surface for
    src.utils.util.read_frames / get_fps / save_videos_from_pil"""

    class VideoFrame:
        def __init__(self, img):
            self.img = img

        @staticmethod
        def from_image(img):
            return _FakeAv.VideoFrame(img)

        def to_image(self):
            return self.img

    class _Stream:
        type = "video"

        def __init__(self, rate):
            self.average_rate = rate
            self.frames = []
            self.options = {}

        def encode(self, frame=None):
            return [] if frame is None else [frame]

    class _Packet:
        def __init__(self, frame):
            self.frame = frame

        def decode(self):
            return [self.frame]

    class _Container:
        def __init__(self, path, mode):
            self.path, self.mode = path, mode
            if mode == "w":
                self.stream = None
            else:
                from PIL import Image

                rec = pickle.load(open(path, "rb"))
                self.stream = _FakeAv._Stream(rec["fps"])
                self.stream.frames = [_FakeAv.VideoFrame(Image.open(io.BytesIO(b)).convert("RGB")) for b in rec["frames"]]

        @property
        def streams(self):
            return [self.stream]

        def add_stream(self, codec, rate=8):
            self.stream = _FakeAv._Stream(rate)
            return self.stream

        def mux(self, frames):
            self.stream.frames.extend(frames)

        def demux(self, stream):
            return [_FakeAv._Packet(f) for f in stream.frames]

        def close(self):
            if self.mode == "w":
                write_fake_video(self.path, [f.img for f in self.stream.frames], self.stream.average_rate)

    @staticmethod
    def open(path, mode="r"):
        return _FakeAv._Container(path, mode)


def write_fake_video(path, pil_frames, fps):
    frames = []
    for im in pil_frames:
        b = io.BytesIO()
        im.save(b, format="PNG")
        frames.append(b.getvalue())
    pickle.dump(dict(fps=fps, frames=frames), open(path, "wb"))


def _stub_modules(TinyVAE, TinyCLIP):
    from PIL import Image

    from humanvid_amd.scheduler import DDIMScheduler

    class VAE(TinyVAE):
        @classmethod
        def from_pretrained(cls, path, **kw):
            assert os.path.isdir(path), path
            return cls()

    class CLIP(TinyCLIP):
        @classmethod
        def from_pretrained(cls, path, **kw):
            assert os.path.isdir(path), path
            return cls()

    class Resize:
        def __init__(self, size):
            self.size = size  # (h, w)

        def __call__(self, img):
            return img.resize((self.size[1], self.size[0]), Image.BILINEAR)

    class ToTensor:
        def __call__(self, img):
            return torch.from_numpy(np.asarray(img.convert("RGB"), dtype=np.float32) / 255.0).permute(2, 0, 1).contiguous()

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    import yaml

    class OmegaConf:
        @staticmethod
        def load(path):
            return _AttrDict(yaml.safe_load(open(path)))

        @staticmethod
        def to_container(cfg, **kw):
            return _plain(cfg)

    mods = {}

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        mods[name] = m
        return m

    mod("av", open=_FakeAv.open, VideoFrame=_FakeAv.VideoFrame)
    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms", Compose=Compose, Resize=Resize, ToTensor=ToTensor)
    sdp = mod("diffusers.pipelines.stable_diffusion", StableDiffusionPipeline=type("StableDiffusionPipeline", (), {}))
    pipes = mod("diffusers.pipelines", stable_diffusion=sdp)
    mod("diffusers", AutoencoderKL=VAE, DDIMScheduler=DDIMScheduler, pipelines=pipes)
    mod("omegaconf", OmegaConf=OmegaConf)
    mod("transformers", CLIPVisionModelWithProjection=CLIP)
    return mods


# ---------------------------------------------------------------------------------------------------------------- test
def test_reference_pose2vid_script_runs_unmodified(tmp_path, monkeypatch):
    text = _script_text()
    if text is None:
        pytest.skip("the reference script is available neither under /root/reference nor in HV_REF_SCRIPT_B64")
    import oracle_torch as O  # weight generators only
    import yaml
    from PIL import Image
    from test_gpu_call_e2e import TinyCLIP, TinyVAE

    root = str(tmp_path)
    # ---- synthetic checkpoints in the layouts the script loads (four levels of width 320: every kernel family, small files)
    cfg = dict(O.SD15_UNET3D_CFG, block_out_channels=(320, 320, 320, 320))
    unet_dir = os.path.join(root, "sd15", "unet")
    os.makedirs(unet_dir)
    sd2d_cfg = dict(in_channels=4, out_channels=4, block_out_channels=[320, 320, 320, 320], layers_per_block=2,
                    cross_attention_dim=768, attention_head_dim=8, norm_num_groups=32, norm_eps=1e-5,
                    down_block_types=["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"],
                    up_block_types=["UpBlock2D"] + ["CrossAttnUpBlock2D"] * 3)
    json.dump(sd2d_cfg, open(os.path.join(unet_dir, "config.json"), "w"))
    sd3 = O.make_unet3d_weights(cfg, seed=31)
    ref_sd = O.make_reference_net_weights(cfg, seed=32)
    torch.save({k: v for k, v in sd3.items() if "motion_modules" not in k}, os.path.join(unet_dir, "diffusion_pytorch_model.bin"))
    torch.save({k: v for k, v in sd3.items() if "motion_modules" in k}, os.path.join(root, "mm.pth"))
    torch.save(sd3, os.path.join(root, "denoising_unet.pth"))
    torch.save(ref_sd, os.path.join(root, "reference_unet.pth"))
    torch.save(O.make_pose_guider_weights(), os.path.join(root, "pose_guider.pth"))
    torch.save(O.make_camera_encoder_weights(), os.path.join(root, "camera_encoder.pth"))
    for d in ("vae", "image_encoder"):
        os.makedirs(os.path.join(root, d))
    # ---- the two YAML files (stage2.yaml / inference_v2.yaml shapes)
    mmk = dict(cfg["motion_module_kwargs"])
    infer = dict(
        unet_additional_kwargs=dict(use_inflated_groupnorm=True, unet_use_cross_frame_attention=False,
                                    unet_use_temporal_attention=False, use_motion_module=True,
                                    motion_module_resolutions=[1, 2, 4, 8], motion_module_mid_block=True,
                                    motion_module_decoder_only=False, motion_module_type="Vanilla",
                                    motion_module_kwargs=dict(mmk, attention_block_types=list(mmk["attention_block_types"]),
                                                              temporal_attention_dim_div=1)),
        noise_scheduler_kwargs=dict(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=False, steps_offset=1,
                                    prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing"),
        sampler="DDIM",
        pose_encoder_kwargs=dict(downscale_factor=8, channels=[320], nums_rb=2, cin=384, ksize=1, sk=True, use_conv=False,
                                 compression_factor=1, temporal_attention_nhead=8, attention_block_types=["Temporal_Self"],
                                 temporal_position_encoding=True, temporal_position_encoding_max_len=24))
    yaml.safe_dump(infer, open(os.path.join(root, "inference.yaml"), "w"))
    ref_root, pose_root = os.path.join(root, "refs"), os.path.join(root, "poses")
    os.makedirs(ref_root), os.makedirs(pose_root)
    rng = np.random.default_rng(7)
    Wd, Ht, frames = 128, 256, 8  # portrait: latent 32 x 16, eight tokens at level 3
    Image.fromarray(rng.integers(0, 255, (Ht, Wd, 3), dtype=np.uint8)).save(os.path.join(ref_root, "clip_a.png"))
    write_fake_video(os.path.join(pose_root, "clip_a.png"),
                     [Image.fromarray(rng.integers(0, 255, (Ht, Wd, 3), dtype=np.uint8)) for _ in range(frames)], 24)
    run_cfg = dict(pretrained_base_model_path=os.path.join(root, "sd15"), pretrained_vae_path=os.path.join(root, "vae"),
                   image_encoder_path=os.path.join(root, "image_encoder"), denoising_unet_path=os.path.join(root, "denoising_unet.pth"),
                   reference_unet_path=os.path.join(root, "reference_unet.pth"), pose_guider_path=os.path.join(root, "pose_guider.pth"),
                   camera_pose_encoder_path=os.path.join(root, "camera_encoder.pth"), motion_module_path=os.path.join(root, "mm.pth"),
                   inference_config=os.path.join(root, "inference.yaml"), weight_dtype="fp16", test_cases={ref_root: [pose_root]})
    yaml.safe_dump(run_cfg, open(os.path.join(root, "stage2_test.yaml"), "w"))

    # ---- run the script as __main__, unmodified
    for name, m in _stub_modules(TinyVAE, TinyCLIP).items():
        monkeypatch.setitem(sys.modules, name, m)
    for name in [n for n in sys.modules if n == "src" or n.startswith("src.") or n.startswith("configs")]:
        monkeypatch.delitem(sys.modules, name)  # fresh import of the drop-in surface with the stubs in place
    script = os.path.join(root, "pose2vid.py")  # a scratch copy OUTSIDE the repository, for runpy
    open(script, "w").write(text)
    monkeypatch.chdir(root)
    monkeypatch.setattr(sys, "argv", ["pose2vid.py", "--config", os.path.join(root, "stage2_test.yaml"), "-W", str(Wd), "-H", str(Ht),
                                      "-L", "4", "--steps", "2", "--cfg", "3.5"])
    np.random.seed(0)
    runpy.run_path(script, run_name="__main__")

    # ---- what the script wrote: output/<config>/<date>/<time>-<W>x<H>/{grid, output_} videos through save_videos_grid
    outs = []
    for d, _, fs in os.walk(os.path.join(root, "output")):
        outs += [os.path.join(d, f) for f in fs if f.endswith(".mp4")]
    assert len(outs) == 2, outs
    solo = [p for p in outs if os.path.basename(p).startswith("output_")]
    assert len(solo) == 1
    rec = pickle.load(open(solo[0], "rb"))
    assert len(rec["frames"]) == 4  # seq_len = L frames
    arr = np.stack([np.asarray(Image.open(io.BytesIO(b))) for b in rec["frames"]])
    assert arr.shape == (4, Ht, Wd, 3) and arr.std() > 1.0  # a real, non-constant decode
    from humanvid_amd import lib as hvlib

    assert hvlib._LIB is not None and os.path.basename(hvlib.LIB_PATH) == "libhumanvid_hip.so"  # the native path ran
    print(f"scripts/pose2vid.py ran unmodified: {len(text.splitlines())} lines, wrote {[os.path.relpath(p, root) for p in outs]}, "
          f"output frames {arr.shape}, mean {arr.mean():.1f}, std {arr.std():.1f}")
