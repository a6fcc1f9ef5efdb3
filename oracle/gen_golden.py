#!/usr/bin/env python3
"""Pin the oracle against the reference's own code and (re)generate tests/golden/*.

TEST INFRASTRUCTURE ONLY -- runs in the build container, where /root/reference is mounted:

    python oracle/gen_golden.py            # check + write fixtures
    python oracle/gen_golden.py --check    # check only (tests/test_host_logic.py runs it when /root/reference is present)

It imports the reference's model files *verbatim* from /root/reference (package name `src`,
which is why this must be its own process: the repo's drop-in surface is also called `src`)
on top of oracle/refshim (stand-ins for the un-vendored diffusers==0.24.0), loads the
oracle's synthetic weights into the reference modules with strict=True (state-dict grammar
check), runs both on the same seeded inputs and asserts agreement.  The fixtures it writes
are what the GPU-side tests consume, since /root/reference does not exist on the GPU box.
"""
import argparse
import importlib.machinery
import importlib.util
import io
import json
import os
import sys
import zipfile
import contextlib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("HV_REFERENCE_ROOT", "/root/reference")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    if not os.path.isdir(REF):
        print("reference tree not present; nothing to do")
        return 2

    # keep the repo root off sys.path so `import src` resolves to the reference
    sys.path = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
    sys.path.insert(0, REF)
    shim = _load("hv_refshim", os.path.join(REPO, "oracle", "refshim", "shim.py"))
    shim.install()
    O = _load("hv_oracle_torch", os.path.join(REPO, "oracle", "oracle_torch.py"))

    import types
    import numpy as np
    import torch

    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    out_dir = os.path.join(REPO, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    report = {}

    def save(name, **arrs):
        if args.check:
            return
        np.savez_compressed(os.path.join(out_dir, name), **{k: np.asarray(v) for k, v in arrs.items()})

    def close(a, b, tol, what):
        err = (a - b).abs().max().item()
        ref = b.abs().max().item()
        report[what] = dict(max_abs_err=err, ref_max=ref)
        assert err <= tol * max(1.0, ref), f"{what}: {err} vs tol {tol} (ref max {ref})"
        print(f"  ok {what}: max|diff|={err:.3e} (ref max {ref:.3f})")

    # ------------------------------------------------------------------ UNet3D, read mode
    from src.models.unet_3d import UNet3DConditionModel
    from src.models.mutual_self_attention import ReferenceAttentionControl
    from src.models.attention import TemporalBasicTransformerBlock

    cfg = O.tiny_unet3d_cfg()
    sd = O.make_unet3d_weights(cfg, seed=0)
    kw = dict(cfg)
    kw.update(use_inflated_groupnorm=True, unet_use_cross_frame_attention=False, unet_use_temporal_attention=False,
              motion_module_type="Vanilla")
    kw["motion_module_kwargs"] = dict(cfg["motion_module_kwargs"], temporal_attention_dim_div=1)
    with contextlib.redirect_stdout(io.StringIO()):
        ref_unet = UNet3DConditionModel(**kw).eval()
    missing, unexpected = ref_unet.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    print("state-dict grammar (tiny): %d tensors, strict load ok" % len(sd))

    b, f, hh, ww = 2, 4, 8, 8
    g = torch.Generator().manual_seed(42)
    sample = torch.randn(1, 4, f, hh, ww, generator=g).repeat(2, 1, 1, 1, 1)
    ehs = torch.cat([torch.zeros(1, 1, 768), torch.randn(1, 1, 768, generator=g)])
    pose = torch.randn(1, 320, f, hh, ww, generator=g).repeat(2, 1, 1, 1, 1) * 0.5
    locs = O.transformer_locations(cfg)
    banks = {}
    for p in locs:
        c = sd[p + ".norm.weight"].numel()
        n_tok = hh * ww if c == 320 else (hh // 2) * (ww // 2)
        banks[p] = torch.randn(2, n_tok, c, generator=g).half().float()
    t = torch.tensor(601)

    reader = ReferenceAttentionControl(ref_unet, do_classifier_free_guidance=True, mode="read", batch_size=1,
                                       fusion_blocks="full")
    blocks = {n: m for n, m in ref_unet.named_modules() if isinstance(m, TemporalBasicTransformerBlock)}
    # the reader's pairing order (sorted by -hidden size, stable) must equal transformer_locations()
    order = sorted(blocks.items(), key=lambda kv: -kv[1].norm1.normalized_shape[0])
    assert [n.rsplit(".transformer_blocks.0", 1)[0] for n, _ in order] == locs, "bank pairing order differs"
    for n, m in blocks.items():
        m.bank = [banks[n.rsplit(".transformer_blocks.0", 1)[0]].half()]  # update() casts to fp16

    with contextlib.redirect_stdout(io.StringIO()):
        ref_out = ref_unet(sample, t, encoder_hidden_states=ehs, pose_cond_fea=pose, return_dict=False)[0]
    taps = {}
    ora_out = O.unet3d_forward(sd, cfg, sample, t, ehs, pose, banks, do_cfg=True, taps=taps)
    close(ora_out, ref_out, 2e-5, "unet3d_tiny_readmode_cfg")
    save("unet3d_tiny.npz", sample=sample[:1].numpy(), ehs=ehs[1:].numpy(), pose=pose[:1].numpy(), t=601,
         out=ref_out.numpy().astype(np.float32),
         **{"bank:" + k: v.numpy().astype(np.float16) for k, v in banks.items()},
         **{"tap:" + k: v.numpy().astype(np.float16) for k, v in taps.items() if "down_blocks.0" in k or k == "mid_block"})

    # frame-independence / CFG semantic probe used by the multi-GPU design
    reader.clear()

    # ------------------------------------------------------------------ full-size key manifest
    full = dict(O.SD15_UNET3D_CFG)
    kwf = dict(kw)
    kwf.update(down_block_types=full["down_block_types"], up_block_types=full["up_block_types"],
               block_out_channels=full["block_out_channels"], layers_per_block=2)
    with torch.device("meta"), contextlib.redirect_stdout(io.StringIO()):
        big = UNet3DConditionModel(**kwf)
    manifest = {k: list(v.shape) for k, v in big.state_dict().items()}
    report["sd15_unet3d_tensors"] = len(manifest)
    report["sd15_unet3d_params_M"] = sum(int(np.prod(s)) for k, s in manifest.items() if not k.endswith(".pe")) / 1e6
    if not args.check:
        with open(os.path.join(out_dir, "unet3d_sd15_keys.json"), "w") as fh:
            json.dump(manifest, fh, indent=0, sort_keys=True)
    print("SD-1.5 UNet3D manifest: %d tensors, %.1f M params" % (len(manifest), report["sd15_unet3d_params_M"]))

    # ------------------------------------------------------------------ PoseGuider
    from src.models.pose_guider import PoseGuider

    pg_sd = O.make_pose_guider_weights()
    pg = PoseGuider(320, block_out_channels=(16, 32, 96, 256)).eval()
    pg.load_state_dict(pg_sd, strict=True)
    cond = torch.rand(1, 3, 3, 64, 48, generator=g)
    close(O.pose_guider_forward(pg_sd, cond), pg(cond), 2e-5, "pose_guider")
    save("pose_guider.npz", cond=cond.numpy(), out=pg(cond).numpy())

    # ------------------------------------------------------------------ CameraPoseEncoder
    from src.cameractrl.pose_adaptor import CameraPoseEncoder

    cam_sd = O.make_camera_encoder_weights()
    ck = dict(O.CAMERA_ENCODER_CFG, channels=[320], attention_block_types=["Temporal_Self"], use_conv=False,
              compression_factor=1)
    cam = CameraPoseEncoder(**ck).eval()
    cam.load_state_dict(cam_sd, strict=True)
    pl = torch.randn(1, 6, 5, 64, 48, generator=g)
    close(O.camera_encoder_forward(cam_sd, pl), cam(pl)[0], 2e-5, "camera_pose_encoder")
    save("camera_encoder.npz", plucker=pl.numpy(), out=cam(pl)[0].numpy())

    # ------------------------------------------------------------------ context windows
    from src.pipelines.context import uniform

    win = {}
    for nf in (8, 16, 24, 32, 48, 72):
        for ov in (0, 4):
            ref_w = list(uniform(0, 30, nf, 24, 1, ov))
            assert O.uniform_windows(0, 30, nf, 24, 1, ov) == ref_w
            win[f"{nf}:{ov}"] = ref_w
    for st in range(4):
        assert O.uniform_windows(st, 30, 48, 16, 3, 4) == list(uniform(st, 30, 48, 16, 3, 4))
    if not args.check:
        with open(os.path.join(out_dir, "context_windows.json"), "w") as fh:
            json.dump(win, fh)
    print("  ok context windows")

    # ------------------------------------------------------------------ camera / Pluecker front-end
    # the dataset module imports torchvision / decord / transformers at the top for its Dataset
    # classes; Camera and ray_condition need none of them, so inert stand-ins are enough
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional", "decord",
                 "src.dataset.visualization_utils", "transformers"):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__dict__.update(VideoReader=None, CameraPoseVisualizer=None, visualize_camera_pose=None, to_image=None,
                          pca_visualize=None, CLIPImageProcessor=None)
        sys.modules[name] = m
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    from src.dataset.dance_image_h_v_camera import Camera, ray_condition

    zf = zipfile.ZipFile(os.path.join(REF, "data", "test_set", "camera_test_set.zip"))
    names = sorted(n for n in zf.namelist() if n.endswith(".txt"))
    rows = [[float(x) for x in ln.split()] for ln in zf.read(names[0]).decode().strip().splitlines()][:7]
    img_size = (48, 64)  # (W, H)
    cams = [Camera(r, "test", img_size) for r in rows]
    K = np.asarray([[c.fx * img_size[0], c.fy * img_size[1], c.cx * img_size[0], c.cy * img_size[1]] for c in cams[1:]],
                   dtype=np.float32)
    abs2rel = np.eye(4) @ cams[0].w2c_mat
    c2w = np.array([np.eye(4)] + [abs2rel @ c.c2w_mat for c in cams[1:]], dtype=np.float32)[1:]
    ref_pl = ray_condition(torch.as_tensor(K)[None], torch.as_tensor(c2w)[None], img_size[1], img_size[0], "cpu")
    ref_pl = ref_pl[0].permute(0, 3, 1, 2).contiguous()[None]
    close(O.plucker_from_entries(rows, img_size), ref_pl, 1e-5, "plucker_camera_test_set")
    save("plucker.npz", rows=np.asarray(rows), img_size=np.asarray(img_size), out=ref_pl.numpy(), source_file=names[0])

    # ------------------------------------------------------------------ DDIM (no reference available: restated only)
    sch = O.DDIM()
    sch.set_timesteps(30)
    ts = sch.timesteps.tolist()
    assert ts[:3] == [999, 966, 932] and ts[-1] == 32 and len(ts) == 30, ts
    assert abs(float(sch.alphas_cumprod[-1])) < 1e-10  # zero terminal SNR
    report["ddim_timesteps_30"] = ts

    if not args.check:
        with open(os.path.join(out_dir, "oracle_pin_report.json"), "w") as fh:
            json.dump(report, fh, indent=1)
    print("oracle pinned against the reference: all checks passed")
    return 0


if __name__ == "__main__":
    sys.exit(main())
