"""Host-side boundary of the drop-in surface (no GPU): checkpoint loader, script helpers, module paths, bench pricing.

 * UNet3DConditionModel.from_pretrained_2d (/root/reference/src/models/unet_3d.py:579-670): a tmp SD-style directory
   (config.json + 2-D safetensors) and a motion-module checkpoint are merged exactly as the reference does -- config
   overlay of the 3-D block types, unet_additional_kwargs, strict=False merge, mm_zero_proj_out, the three error paths.
 * src.utils.util helpers (/root/reference/src/utils/util.py): grid layout, .gif writing, seeding.
 * every module path scripts/pose2vid.py / scripts/pose2img.py import from `src.*` resolves to the native package.
"""
import importlib
import json
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import oracle_torch as O  # noqa: E402


def _tiny_kwargs():
    cfg = O.tiny_unet3d_cfg()
    return cfg, dict(use_inflated_groupnorm=True, unet_use_cross_frame_attention=False, unet_use_temporal_attention=False,
                     use_motion_module=True, motion_module_resolutions=(1, 2, 4, 8), motion_module_mid_block=True,
                     motion_module_type="Vanilla", motion_module_kwargs=dict(cfg["motion_module_kwargs"]))


def test_from_pretrained_2d_round_trip(tmp_path):
    from safetensors.torch import save_file

    from humanvid_amd.unet3d import UNet3DConditionModel

    cfg, extra = _tiny_kwargs()
    sd = O.make_unet3d_weights(cfg, seed=3)
    spatial = {k: v.contiguous() for k, v in sd.items() if "motion_modules" not in k}
    motion = {k: v.contiguous() for k, v in sd.items() if "motion_modules" in k}
    assert spatial and motion
    unet_dir = tmp_path / "sd" / "unet"
    unet_dir.mkdir(parents=True)
    # an SD-1.5 style 2-D config: 2-D block names (overwritten by the loader), geometry keys the loader keeps
    conf = dict(_class_name="UNet2DConditionModel", in_channels=4, out_channels=4, layers_per_block=cfg["layers_per_block"],
                block_out_channels=list(cfg["block_out_channels"]), cross_attention_dim=768, attention_head_dim=8,
                down_block_types=["CrossAttnDownBlock2D", "DownBlock2D"], up_block_types=["UpBlock2D", "CrossAttnUpBlock2D"],
                norm_num_groups=32, some_unknown_key=1)
    (unet_dir / "config.json").write_text(json.dumps(conf))
    save_file(spatial, str(unet_dir / "diffusion_pytorch_model.safetensors"))
    mm = tmp_path / "mm.pth"
    torch.save(motion, mm)
    # the loader hard-codes the 4-level 3-D block types (unet_3d.py:603-616); a 2-level checkpoint therefore needs them
    # overridden -- which `unet_additional_kwargs` allows, as it is applied last in from_config
    extra2 = dict(extra, down_block_types=cfg["down_block_types"], up_block_types=cfg["up_block_types"])
    net = UNet3DConditionModel.from_pretrained_2d(str(tmp_path / "sd"), str(mm), subfolder="unet",
                                                  unet_additional_kwargs=extra2)
    got = net.state_dict()
    assert set(got) == set(sd)
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    assert net.config["cross_attention_dim"] == 768 and net.config["use_motion_module"] is True
    assert "some_unknown_key" not in net.config

    # mm_zero_proj_out: proj_out tensors of the motion checkpoint are dropped -> those parameters keep their zero init
    net0 = UNet3DConditionModel.from_pretrained_2d(str(tmp_path / "sd"), str(mm), subfolder="unet",
                                                   unet_additional_kwargs=extra2, mm_zero_proj_out=True)
    po = [k for k in sd if "motion_modules" in k and "proj_out" in k]
    assert po and all(float(net0.state_dict()[k].abs().max()) == 0.0 for k in po)
    other = [k for k in motion if "proj_out" not in k]
    assert all(torch.equal(net0.state_dict()[k], sd[k]) for k in other)

    # .safetensors motion module, missing motion file (silently skipped, as the reference does), error paths
    save_file(motion, str(tmp_path / "mm.safetensors"))
    net1 = UNet3DConditionModel.from_pretrained_2d(str(tmp_path / "sd"), str(tmp_path / "mm.safetensors"), subfolder="unet",
                                                   unet_additional_kwargs=extra2)
    assert all(torch.equal(net1.state_dict()[k], sd[k]) for k in sd)
    net2 = UNet3DConditionModel.from_pretrained_2d(str(tmp_path / "sd"), str(tmp_path / "absent.ckpt"), subfolder="unet",
                                                   unet_additional_kwargs=extra2)
    assert all(torch.equal(net2.state_dict()[k], sd[k]) for k in spatial)
    (tmp_path / "mm.bad").write_bytes(b"x")
    with pytest.raises(RuntimeError):
        UNet3DConditionModel.from_pretrained_2d(str(tmp_path / "sd"), str(tmp_path / "mm.bad"), subfolder="unet",
                                                unet_additional_kwargs=extra2)
    with pytest.raises(RuntimeError):
        UNet3DConditionModel.from_pretrained_2d(str(tmp_path / "nowhere"), str(mm), unet_additional_kwargs=extra2)
    os.remove(unet_dir / "diffusion_pytorch_model.safetensors")
    with pytest.raises(FileNotFoundError):
        UNet3DConditionModel.from_pretrained_2d(str(tmp_path / "sd"), str(mm), subfolder="unet",
                                                unet_additional_kwargs=extra2)
    torch.save(spatial, unet_dir / "diffusion_pytorch_model.bin")  # the .bin fallback
    net3 = UNet3DConditionModel.from_pretrained_2d(str(tmp_path / "sd"), str(mm), subfolder="unet",
                                                   unet_additional_kwargs=extra2)
    assert all(torch.equal(net3.state_dict()[k], sd[k]) for k in sd)
    # a container never computes on the host
    with pytest.raises(RuntimeError):
        net3(torch.zeros(2, 4, 2, 8, 8), 1, torch.zeros(2, 1, 768))


def test_script_module_paths_resolve_to_the_native_package():
    wanted = {
        "src.pipelines.pipeline_pose2vid_long": ["Pose2VideoPipeline", "Pose2VideoPipelineOutput"],
        "src.pipelines.pipeline_pose2vid": ["Pose2VideoPipeline"],
        "src.pipelines.pipeline_pose2img": ["Pose2ImagePipeline", "Pose2ImagePipelineOutput"],
        "src.pipelines.context": ["get_context_scheduler", "uniform", "ordered_halving"],
        "src.models.unet_3d": ["UNet3DConditionModel"],
        "src.models.unet_2d_condition": ["UNet2DConditionModel"],
        "src.models.pose_guider": ["PoseGuider"],
        "src.models.mutual_self_attention": ["ReferenceAttentionControl"],
        "src.cameractrl.pose_adaptor": ["CameraPoseEncoder"],
        "src.dataset.dance_image_h_v_camera": ["Camera", "ray_condition"],
        "src.utils.util": ["get_fps", "read_frames", "save_videos_grid", "save_image_grid", "seed_everything",
                           "save_checkpoint", "delete_additional_ckpt", "show_image_grid"],  # (train_stage_1.py:42, train_stage_2.py:45)
        "configs.prompts.test_cases": ["TestCasesDict"],
    }
    for mod, names in wanted.items():
        m = importlib.import_module(mod)
        for n in names:
            obj = getattr(m, n)
            if callable(obj) and hasattr(obj, "__module__"):
                assert obj.__module__.startswith(("humanvid_amd", "configs")), (mod, n, obj.__module__)
    import inspect

    from src.pipelines.pipeline_pose2img import Pose2ImagePipeline
    from src.pipelines.pipeline_pose2vid import Pose2VideoPipeline as Short
    from src.pipelines.pipeline_pose2vid_long import Pose2VideoPipeline as Long

    # argument order of the three __call__ variants (pipeline_pose2vid_long.py:340-364, pipeline_pose2vid.py:285-301,
    # pipeline_pose2img.py:195-211)
    assert list(inspect.signature(Long.__call__).parameters)[1:9] == [
        "ref_image", "pose_images", "camera_embedding", "width", "height", "video_length", "num_inference_steps", "guidance_scale"]
    assert list(inspect.signature(Short.__call__).parameters)[1:8] == [
        "ref_image", "pose_images", "width", "height", "video_length", "num_inference_steps", "guidance_scale"]
    assert list(inspect.signature(Pose2ImagePipeline.__call__).parameters)[1:8] == [
        "ref_image", "pose_image", "camera_embedding", "width", "height", "num_inference_steps", "guidance_scale"]
    assert list(inspect.signature(Short.__init__).parameters)[1:7] == [
        "vae", "image_encoder", "reference_unet", "denoising_unet", "pose_guider", "scheduler"]


def test_util_grid_and_gif(tmp_path):
    from PIL import Image

    from src.utils.util import make_grid, save_image_grid, save_videos_grid, seed_everything

    seed_everything(7)
    a = torch.rand(2)
    seed_everything(7)
    assert torch.equal(a, torch.rand(2))
    vids = torch.rand(3, 3, 4, 8, 6)  # b c t h w, portrait -> n_rows columns
    g = make_grid(vids[:, :, 0], nrow=4)
    assert g.shape == (3, 8 + 4, 3 * (6 + 2) + 2)  # one row of three, padding 2
    assert torch.equal(g[:, 2:10, 2:8], vids[0, :, 0]) and torch.equal(g[:, 2:10, 10:16], vids[1, :, 0])
    assert float(g[:, 0].abs().max()) == 0.0
    assert make_grid(vids[:1, :, 0]).shape == (3, 8, 6)  # a single image is not padded (torchvision behaviour)
    wide = torch.rand(3, 3, 2, 6, 8)  # landscape: two per row -> 2 rows
    out = tmp_path / "o" / "grid.gif"
    save_videos_grid(wide, str(out), n_rows=4, fps=4)
    im = Image.open(out)
    assert im.n_frames == 2 and im.size == (2 * (8 + 2) + 2, 2 * (6 + 2) + 2)
    save_image_grid(vids[:, :, 0], str(tmp_path / "o" / "grid.png"))
    assert Image.open(tmp_path / "o" / "grid.png").size == (3 * 8 + 2, 12)
    with pytest.raises(ValueError):
        save_videos_grid(wide, str(tmp_path / "o" / "x.avi"))


def test_util_checkpoint_bookkeeping(tmp_path):
    """save_checkpoint / delete_additional_ckpt as the reference's training scripts use them (src/utils/util.py:17-45, 66-79):
    `<prefix>-<n>.pth` files pruned to total_limit, the motion_module filter, `checkpoint-<n>` directories pruned to num_keep"""
    from src.utils.util import delete_additional_ckpt, save_checkpoint

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = torch.nn.Linear(2, 2)
            self.motion_modules = torch.nn.ModuleList([torch.nn.Linear(2, 2)])

    net = Net()
    for step in (10, 20, 30, 40):
        save_checkpoint(net, str(tmp_path), "denoising_unet", step, total_limit=3)
    assert sorted(os.listdir(tmp_path)) == ["denoising_unet-20.pth", "denoising_unet-30.pth", "denoising_unet-40.pth"]
    assert set(torch.load(tmp_path / "denoising_unet-40.pth")) == set(net.state_dict())
    save_checkpoint(net, str(tmp_path), "motion_module", 5)
    assert set(torch.load(tmp_path / "motion_module-5.pth")) == {"motion_modules.0.weight", "motion_modules.0.bias"}
    for n in (100, 200, 300):
        os.makedirs(tmp_path / f"checkpoint-{n}")
    delete_additional_ckpt(str(tmp_path), 1)
    assert sorted(d for d in os.listdir(tmp_path) if d.startswith("checkpoint-")) == ["checkpoint-300"]


def test_config5_windows_and_bench_pricing():
    """BASELINE.json configs[4]: 48 frames -> three windows of 24 per step; bench.py prices launches from
    the profile keys the library emits."""
    from humanvid_amd.scheduler import get_context_scheduler

    win = list(get_context_scheduler("uniform")(0, 30, 48, 24, 1, 4))
    assert win == [list(range(24)), list(range(20, 44)), list(range(40, 48)) + list(range(16))]
    import bench

    fl, by = bench.price_launch("hv_gemm_glds_kernel<32,3,128,4> | M=294912 N=960 K=320 geglu=0 res=0 yt=320 f32=0 x2=0")
    assert fl == 2.0 * 294912 * 960 * 320 and by == 2.0 * (294912 * 320 + 960 * 320 + 294912 * 960)
    fl, _ = bench.price_launch("hv_attention_kernel<40,2> | n=48 heads=8 D=40 Lq=6144 L1=6144 L2=6144 bank=1")
    assert abs(fl - 3478923509760.0) < 1.0  # the figure of the round-1 attention probe
    fl, by = bench.price_launch("hv_conv3x3_kernel<16,0,1,128> | n=48 Hs=96 Ws=64 Ho=96 Wo=64 Cin=320 Cout=320 gn=1 res=1")
    assert fl == 2.0 * 9 * 320 * 320 * 48 * 6144
    assert bench.price_launch("hv_pack_kernel") == (0.0, 0.0)
    fl, by = bench.price_launch("hv_affine_apply_kernel | rows=73728 C=640 C2=320 act=2")
    assert fl == 0.0 and by == 4.0 * 73728 * 960 and bench.price_launch_rw("hv_affine_apply_kernel | rows=8 C=8 C2=0 act=0") == (128.0, 128.0)
    # a launch note with a field this module cannot parse is priced as nothing, it does not take the bench line down
    assert bench.price_launch("hv_some_future_kernel | rows=96+0 mode=x") == (0.0, 0.0)
