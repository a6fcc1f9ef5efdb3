#!/usr/bin/env python3
"""Golden vectors of the LOOP BODY at the benchmarked size, from the REFERENCE's own modules  --  TEST INFRASTRUCTURE ONLY.

    python oracle/gen_fullsize_steps_golden.py steps3      # two consecutive steps (t = 999, 966) at 24f x 96x64 latents
    python oracle/gen_fullsize_steps_golden.py windows48   # one + one step of a 48-frame clip, three windows per step
    python oracle/gen_fullsize_steps_golden.py edge_t32    # the last step of the 30-step schedule (final_alpha_cumprod)
    python oracle/gen_fullsize_steps_golden.py traj30      # all 30 steps at 24f x 96x64 (2.5 h on 8 cores); keeps the latents
                                                           # after steps 0, 4, 9, ... 29 as fp16 and every step's rms

Runs in the build container (needs /root/reference).  The reference's UNet3DConditionModel (read mode, CFG, seeded fp16
banks), PoseGuider and CameraPoseEncoder are imported verbatim and driven by the statements of the reference's loop body,
src/pipelines/pipeline_pose2vid_long.py:454-563: context windows from the reference's own src/pipelines/context.py,
per-window PoseGuider + CameraPoseEncoder, CFG batch, noise_pred / counter accumulation, guidance, scheduler step.  The
scheduler is oracle_torch.DDIM (diffusers is not installable here: the restatement of DDIMScheduler under
configs/inference/inference_v2.yaml).  Everything is fp32 on the host cores.

Output tests/golden/steps_<case>.npz:  for every executed step i:  "noise_pred<i>" (after guidance, fp16), "latents<i>"
(after scheduler.step, fp32), "counter<i>", "t<i>";  consumed by tests/test_gpu_fullsize_steps.py on the GPU box.
"""
import contextlib
import io
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refenv  # noqa: E402


def main():
    case = sys.argv[1]
    if not refenv.present():
        print("reference tree not present; nothing to do")
        return 2
    shim, O = refenv.enter()
    FC = refenv._load("hv_fullsize_case", os.path.join(refenv.REPO, "tests", "fullsize_case.py"))
    import numpy as np
    import torch
    from einops import rearrange

    torch.set_num_threads(int(os.environ.get("HV_THREADS", os.cpu_count())))
    torch.set_grad_enabled(False)
    from src.models.unet_3d import UNet3DConditionModel
    from src.models.mutual_self_attention import ReferenceAttentionControl
    from src.models.attention import TemporalBasicTransformerBlock
    from src.models.pose_guider import PoseGuider
    from src.cameractrl.pose_adaptor import CameraPoseEncoder
    from src.pipelines.context import get_context_scheduler

    geo = FC.STEP_CASES[case]
    cfg = dict(O.SD15_UNET3D_CFG)
    sd = O.make_unet3d_weights(cfg, seed=FC.WEIGHT_SEED)
    kw = dict(cfg)
    kw.update(use_inflated_groupnorm=True, unet_use_cross_frame_attention=False, unet_use_temporal_attention=False,
              motion_module_type="Vanilla")
    kw["motion_module_kwargs"] = dict(cfg["motion_module_kwargs"], temporal_attention_dim_div=1)
    with contextlib.redirect_stdout(io.StringIO()):
        unet = UNet3DConditionModel(**kw).eval()
    unet.load_state_dict(sd, strict=True)
    locs = O.transformer_locations(cfg)
    latents, pose_cond_tensor, camera_embedding, clip, banks = FC.make_step_inputs(
        case, locs, lambda p: sd[p + ".norm.weight"].numel())
    del sd
    ReferenceAttentionControl(unet, do_classifier_free_guidance=True, mode="read", batch_size=1, fusion_blocks="full")
    for n, m in unet.named_modules():
        if isinstance(m, TemporalBasicTransformerBlock):
            m.bank = [banks[n.rsplit(".transformer_blocks.0", 1)[0]].half()]  # update() casts to fp16 (:338)
    pose_guider = PoseGuider(**O.POSE_GUIDER_CFG).eval()
    pose_guider.load_state_dict(O.make_pose_guider_weights(), strict=True)
    ck = dict(O.CAMERA_ENCODER_CFG, channels=[320], attention_block_types=["Temporal_Self"], use_conv=False,
              compression_factor=1)
    camera_pose_encoder = CameraPoseEncoder(**ck).eval()
    camera_pose_encoder.load_state_dict(O.make_camera_encoder_weights(), strict=True)

    guidance_scale = FC.GUIDANCE
    num_inference_steps = geo["num_inference_steps"]
    scheduler = O.DDIM()
    scheduler.set_timesteps(num_inference_steps)
    timesteps = scheduler.timesteps
    encoder_hidden_states = torch.cat([torch.zeros_like(clip), clip], dim=0)  # :387-392 (zero uncond embedding)
    context_scheduler = get_context_scheduler("uniform")
    context_frames, context_stride, context_overlap, context_batch_size = 24, 1, 4, 1
    arrs = dict(F=geo["F"], h=geo["h"], w=geo["w"], steps=np.array(geo["steps"]), guidance=guidance_scale,
                num_inference_steps=num_inference_steps)
    for i in geo["steps"]:
        t = timesteps[i]
        t0 = time.time()
        # ---- pipeline_pose2vid_long.py:455-563, statement by statement (reference-net write pass at i == 0 replaced by the
        #      seeded banks installed above)
        noise_pred = torch.zeros((latents.shape[0] * 2, *latents.shape[1:]), dtype=latents.dtype)
        counter = torch.zeros((1, 1, latents.shape[2], 1, 1), dtype=latents.dtype)
        context_queue = list(context_scheduler(0, num_inference_steps, latents.shape[2], context_frames, context_stride,
                                               context_overlap))
        num_context_batches = -(-len(context_queue) // context_batch_size)
        global_context = [context_queue[k * context_batch_size:(k + 1) * context_batch_size]
                          for k in range(num_context_batches)]
        for context in global_context:
            latent_model_input = torch.cat([latents[:, :, c] for c in context]).repeat(2, 1, 1, 1, 1)
            b, c_, f, h, w = latent_model_input.shape
            cur_pose_cond = torch.cat([pose_cond_tensor[:, :, c] for c in context])
            pose_fea = pose_guider(cur_pose_cond)
            latent_pose_input = pose_fea.repeat(2, 1, 1, 1, 1)
            cur_camera_embedding = torch.cat([camera_embedding[:, :, c] for c in context])
            camera_b = cur_camera_embedding.shape[0]
            cur_camera_embedding = camera_pose_encoder(cur_camera_embedding)[0]
            cur_camera_embedding = rearrange(cur_camera_embedding, "(b f) c h w -> b c f h w", b=camera_b)
            latent_camera_input = cur_camera_embedding.repeat(2, 1, 1, 1, 1)
            with contextlib.redirect_stdout(io.StringIO()):
                pred = unet(latent_model_input, t, encoder_hidden_states=encoder_hidden_states[:b],
                            pose_cond_fea=latent_pose_input + latent_camera_input, return_dict=False)[0]
            for j, c in enumerate(context):
                noise_pred[:, :, c] = noise_pred[:, :, c] + pred
                counter[:, :, c] = counter[:, :, c] + 1
        noise_pred_uncond, noise_pred_text = (noise_pred / counter).chunk(2)
        noise_pred = noise_pred_uncond + guidance_scale * (noise_pred_text - noise_pred_uncond)
        if geo["steps"][0] != 0 and i == geo["steps"][0]:
            arrs["latents_in"] = latents.numpy().copy()
        latents = scheduler.step(noise_pred, int(t), latents)
        assert torch.isfinite(latents).all()
        keep = geo.get("keep")
        if keep is None:
            arrs[f"noise_pred{i}"] = noise_pred.half().numpy()
            arrs[f"latents{i}"] = latents.numpy().copy()
            arrs[f"counter{i}"] = counter.reshape(-1).numpy().copy()
        else:  # a long trajectory: the latents of a few steps as fp16, rms of every step; written as it goes
            arrs[f"latent_rms{i}"] = float(latents.pow(2).mean().sqrt())
            arrs[f"noise_rms{i}"] = float(noise_pred.pow(2).mean().sqrt())
            if i in keep:
                arrs[f"latents{i}"] = latents.half().numpy()
                np.savez_compressed(os.path.join(refenv.REPO, "tests", "golden", f"steps_{case}.partial.npz"), **arrs)
        arrs[f"t{i}"] = int(t)
        print(f"[{case}] step {i} t={int(t)} windows={len(context_queue)} {time.time() - t0:.0f}s  "
              f"noise rms {noise_pred.pow(2).mean().sqrt():.4f} latent rms {latents.pow(2).mean().sqrt():.4f}", flush=True)
    out = os.path.join(refenv.REPO, "tests", "golden", f"steps_{case}.npz")
    np.savez_compressed(out, **arrs)
    print("wrote", out, os.path.getsize(out) >> 10, "KiB")
    return 0


if __name__ == "__main__":
    sys.exit(main())
