#!/usr/bin/env python3
"""Full-size golden vectors + CPU baseline from the REFERENCE's own code  --  TEST INFRASTRUCTURE ONLY.

    python oracle/gen_fullsize_golden.py config3          # 24f x 768x512  (BASELINE.json configs[2])
    python oracle/gen_fullsize_golden.py config2          # 16f x 512x512  (configs[1])
    HV_TIMED_REPEATS=0 python oracle/gen_fullsize_golden.py config5   # one 24f x 1024x576 window of configs[4], golden only

Runs in the build container (needs /root/reference).  It imports the reference's UNet3DConditionModel,
ReferenceAttentionControl, PoseGuider and CameraPoseEncoder *verbatim* (on top of oracle/refshim),
loads oracle_torch.make_unet3d_weights(SD15_UNET3D_CFG) with strict=True, installs seeded fp16 banks
in read mode and executes one CFG forward of src/models/unet_3d.py:397-577 at the full benchmark size
in fp32 on the host cores.  Forward hooks on the reference modules record, for every resnet /
spatial transformer / motion module of the down path, the mid block and every (block, layer) of the
up path:  a [2 images x pixel grid x all channels] slice (fp16) and the per-image rms of the whole
activation.  Output + taps go to tests/golden/unet3d_<case>.npz; tests/test_gpu_fullsize_parity.py
compares the native path against them on the GPU box.

The same run is the CPU baseline SURVEY.md 8(d) specifies: the second, steady-state execution of one
denoising step (PoseGuider + CameraPoseEncoder + UNet3D, as pipeline_pose2vid_long.py:526-548
recomputes them every step) is timed with torch.set_num_threads(<cores of this container>) and
written to tests/golden/cpu_reference_<case>.json.
"""
import contextlib
import io
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refenv  # noqa: E402


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "config3"
    timed_repeats = int(os.environ.get("HV_TIMED_REPEATS", "1"))
    if not refenv.present():
        print("reference tree not present; nothing to do")
        return 2
    shim, O = refenv.enter()
    FC = refenv._load("hv_fullsize_case", os.path.join(refenv.REPO, "tests", "fullsize_case.py"))

    import numpy as np
    import torch

    ncpu = os.cpu_count()
    torch.set_num_threads(ncpu)
    torch.set_grad_enabled(False)

    from src.models.unet_3d import UNet3DConditionModel
    from src.models.mutual_self_attention import ReferenceAttentionControl
    from src.models.attention import TemporalBasicTransformerBlock
    from src.models.pose_guider import PoseGuider
    from src.cameractrl.pose_adaptor import CameraPoseEncoder

    cfg = dict(O.SD15_UNET3D_CFG)
    geo = FC.CASES[case]
    F, h, w = geo["F"], geo["h"], geo["w"]
    t0 = time.time()
    sd = O.make_unet3d_weights(cfg, seed=FC.WEIGHT_SEED)
    kw = dict(cfg)
    kw.update(use_inflated_groupnorm=True, unet_use_cross_frame_attention=False, unet_use_temporal_attention=False,
              motion_module_type="Vanilla")
    kw["motion_module_kwargs"] = dict(cfg["motion_module_kwargs"], temporal_attention_dim_div=1)
    with contextlib.redirect_stdout(io.StringIO()):
        unet = UNet3DConditionModel(**kw).eval()
    missing, unexpected = unet.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    print(f"[{case}] reference UNet3D built, {len(sd)} tensors, {time.time() - t0:.0f}s", flush=True)

    locs = O.transformer_locations(cfg)
    sample, ehs, pose, banks = FC.make_inputs(case, locs, lambda p: sd[p + ".norm.weight"].numel())
    del sd
    ReferenceAttentionControl(unet, do_classifier_free_guidance=True, mode="read", batch_size=1, fusion_blocks="full")
    blocks = {n: m for n, m in unet.named_modules() if isinstance(m, TemporalBasicTransformerBlock)}
    for n, m in blocks.items():
        m.bank = [banks[n.rsplit(".transformer_blocks.0", 1)[0]].half()]  # update() casts to fp16 (:338)

    # ---- taps through forward hooks on the reference's own modules
    taps, stats = {}, {}
    recording = [True]

    def hook(name):
        def fn(mod, args, out):
            if not recording[0]:
                return
            x = out.sample if hasattr(out, "sample") else (out[0] if isinstance(out, (tuple, list)) else out)
            taps[name] = FC.slice_ncfhw(x, F).half().numpy()
            stats[name] = FC.rms_ncfhw(x).numpy()
        return fn

    mods = dict(unet.named_modules())
    spec = O.unet3d_spec(cfg)
    for blk in spec["down"]:
        p = blk["prefix"]
        for j in range(len(blk["resnets"])):
            mods[f"{p}.resnets.{j}"].register_forward_hook(hook(f"{p}.resnets.{j}"))
            if blk["attn"]:
                mods[f"{p}.attentions.{j}"].register_forward_hook(hook(f"{p}.attentions.{j}"))
            if blk["motion"]:
                mods[f"{p}.motion_modules.{j}"].register_forward_hook(hook(f"{p}.motion_modules.{j}"))
    mods["mid_block.resnets.1"].register_forward_hook(hook("mid_block"))
    for blk in spec["up"]:
        p = blk["prefix"]
        for j in range(len(blk["resnets"])):
            last = "motion_modules" if blk["motion"] else ("attentions" if blk["attn"] else "resnets")
            mods[f"{p}.{last}.{j}"].register_forward_hook(hook(f"{p}.{j}"))

    t = torch.tensor(FC.TIMESTEP)
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        out = unet(sample, t, encoder_hidden_states=ehs, pose_cond_fea=pose, return_dict=False)[0]
    first = time.time() - t0
    print(f"[{case}] first forward {first:.1f}s, out rms {out.pow(2).mean().sqrt():.4f}", flush=True)
    assert torch.isfinite(out).all()
    recording[0] = False
    arrs = dict(out=out.half().numpy(), t=FC.TIMESTEP, F=F, h=h, w=w)
    arrs.update({"tap:" + k: v for k, v in taps.items()})
    arrs.update({"rms:" + k: v for k, v in stats.items()})
    gold = os.path.join(refenv.REPO, "tests", "golden")
    np.savez_compressed(os.path.join(gold, f"unet3d_{case}.npz"), **arrs)
    print(f"[{case}] wrote unet3d_{case}.npz with {len(taps)} taps", flush=True)

    # ---- CPU baseline: steady-state denoising step of the reference source
    pg = PoseGuider(**O.POSE_GUIDER_CFG).eval()
    pg.load_state_dict(O.make_pose_guider_weights(), strict=True)
    ck = dict(O.CAMERA_ENCODER_CFG, channels=[320], attention_block_types=["Temporal_Self"], use_conv=False,
              compression_factor=1)
    cam = CameraPoseEncoder(**ck).eval()
    cam.load_state_dict(O.make_camera_encoder_weights(), strict=True)
    pose_img = torch.rand(1, 3, F, h * 8, w * 8, generator=torch.Generator().manual_seed(1))
    plucker = torch.randn(1, 6, F, h * 8, w * 8, generator=torch.Generator().manual_seed(3))
    if timed_repeats <= 0:  # golden only (HV_TIMED_REPEATS=0)
        return 0
    times = []
    for _ in range(timed_repeats):
        t0 = time.time()
        pf = pg(pose_img)
        t1 = time.time()
        cf = cam(plucker)[0]
        cf = cf.view(1, F, *cf.shape[1:]).permute(0, 2, 1, 3, 4)
        t2 = time.time()
        cond = (pf + cf).repeat(2, 1, 1, 1, 1)
        with contextlib.redirect_stdout(io.StringIO()):
            unet(sample, t, encoder_hidden_states=ehs, pose_cond_fea=cond, return_dict=False)
        t3 = time.time()
        times.append(dict(pose_guider_s=t1 - t0, camera_encoder_s=t2 - t1, unet_s=t3 - t2, step_s=t3 - t0))
        print(f"[{case}] timed step: {times[-1]}", flush=True)
    best = min(times, key=lambda d: d["step_s"])
    rep = dict(case=case, frames=F, latent=[h, w], kind="reference source (/root/reference) + oracle/refshim, fp32",
               host="build container", cores=ncpu, torch_threads=torch.get_num_threads(),
               first_unet_forward_s=first, steady_state=best, steps_per_s=1.0 / best["step_s"])
    with open(os.path.join(gold, f"cpu_reference_{case}.json"), "w") as fh:
        json.dump(rep, fh, indent=1)
    print(json.dumps(rep))
    return 0


if __name__ == "__main__":
    sys.exit(main())
