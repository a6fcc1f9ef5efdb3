"""Parity of the LOOP BODY at the benchmarked size against the reference's own modules ("parity beyond one
forward").

tests/golden/steps_<case>.npz were produced in the build container by oracle/gen_fullsize_steps_golden.py: the reference's
UNet3DConditionModel (read mode, CFG, seeded fp16 banks), PoseGuider and CameraPoseEncoder, imported verbatim from
/root/reference and driven by the statements of pipeline_pose2vid_long.py:455-563 (context windows from the reference's own
src/pipelines/context.py, noise_pred / counter accumulation, guidance, DDIM step), fp32 on the host cores:

  steps3     two CONSECUTIVE steps of the 30-step schedule at config #3's size (24 frames, 96 x 64 latent, SD-1.5 widths):
             i = 0 is t = 999 -- zero terminal SNR, sqrt(abar_t) = 0, the step the single-forward fixture (t = 499) never saw --
             and i = 1 (t = 966) starts from the native path's OWN step-0 latents: the error of step 0 rides along
  edge_t32   the LAST step (i = 29, t = 32: prev_timestep < 0 -> final_alpha_cumprod) at the same size
  windows48  two steps of a 48-frame clip (context 24, overlap 4: three overlapping windows per step, the geometry of
             config #5) at a 32 x 32 latent: hv_accumulate_window + per-frame counter division with the level-0 kernels

Compared per step: the latents after the scheduler step and the guided noise prediction (recovered from the latent update,
which is linear in it).  Stated tolerances (bf16 storage, fp32 accumulation; measured values are printed):
guided noise prediction NRMSE <= 2e-2 per step (the bound a single forward is held to), latents <= 1e-2 (they carry
sqrt(1 - abar_prev) of the prediction error, less at the early steps), every frame's noise prediction within 1.5 x its bound.
The guided prediction u + 3.5 (c - u) = 3.5 c - 2.5 u weights the two halves' errors with 3.5 and 2.5 while its own norm stays that
of c; the storage-format floor of one forward is 1.44e-2 (oracle/storage_model.py, tests/golden/unet3d_config3_storage.npz
`floor_out`) and the halves' rounding errors are strongly correlated, so the guided one measures 1.58 - 1.85e-2 (round 5 had
loosened this bound to 2.5e-2; round 6 restores 2e-2: every GEMM / norm change since is bit-identical or re-measured under it).
The check that is meant to see a small kernel regression is tests/test_gpu_storage_model.py (per block, bounds 2 - 4e-3)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
sys.path.insert(0, os.path.dirname(__file__))
import fullsize_case as FC  # noqa: E402
import oracle_torch as O  # noqa: E402  (test infrastructure: weight generators only)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL_NOISE, TOL_LAT = 2e-2, 1e-2


def nrmse(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


@pytest.fixture(scope="module")
def pipe_and_chan():
    from humanvid_amd.conditioning import CameraPoseEncoder, PoseGuider
    from humanvid_amd.pipeline import Pose2VideoPipeline
    from humanvid_amd.scheduler import DDIMScheduler
    from humanvid_amd.unet3d import UNet3DConditionModel

    cfg = dict(O.SD15_UNET3D_CFG)
    sd = O.make_unet3d_weights(cfg, seed=FC.WEIGHT_SEED)
    net = UNet3DConditionModel(**dict(cfg, use_inflated_groupnorm=True, unet_use_cross_frame_attention=False,
                                      unet_use_temporal_attention=False, motion_module_type="Vanilla"))
    net.load_state_dict(sd, strict=True)
    chan = {p: sd[p + ".norm.weight"].numel() for p in O.transformer_locations(cfg)}
    del sd
    net = net.to("cuda")
    pg = PoseGuider(**O.POSE_GUIDER_CFG)
    pg.load_state_dict(O.make_pose_guider_weights(), strict=True)
    ck = dict(O.CAMERA_ENCODER_CFG, channels=[320], attention_block_types=["Temporal_Self"], use_conv=False, compression_factor=1)
    cam = CameraPoseEncoder(**ck)
    cam.load_state_dict(O.make_camera_encoder_weights(), strict=True)
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=False, steps_offset=1,
                          prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")
    return Pose2VideoPipeline(None, None, None, net, pg.to("cuda"), cam.to("cuda"), sched), chan


@pytest.mark.parametrize("case", ["windows48", "steps3", "edge_t32"])
def test_native_loop_steps_match_the_reference_at_full_size(pipe_and_chan, case):
    from humanvid_amd.scheduler import fused_step_coefficients

    pipe, chan = pipe_and_chan
    z = np.load(os.path.join(GOLD, f"steps_{case}.npz"))
    steps = [int(s) for s in z["steps"]]
    n_inf = int(z["num_inference_steps"])
    lat, pose, pl, clip, banks = FC.make_step_inputs(case, list(chan), lambda p: chan[p])
    eng = pipe.denoising_unet.engine()
    eng.set_reference_banks({k: v.cuda() for k, v in banks.items()}, do_cfg=True)
    eng._banks_from_modules = lambda: None
    got = []
    x_in = lat.clone()
    pipe.denoise(lat.clone().cuda(), pose.cuda(), pl.cuda(), clip.cuda(), n_inf, FC.GUIDANCE, first_step=steps[0],
                 max_steps=len(steps), callback=lambda i, t, x: got.append((int(t), x.detach().float().cpu().clone())))
    torch.cuda.synchronize()
    assert len(got) == len(steps)
    pipe.scheduler.set_timesteps(n_inf)
    for i, (t, x_out) in zip(steps, got):
        assert t == int(z[f"t{i}"]), (t, int(z[f"t{i}"]))  # the callback receives the scheduler's timestep, as in the reference
        assert torch.isfinite(x_out).all()
        want_lat = torch.from_numpy(z[f"latents{i}"])
        want_noise = torch.from_numpy(z[f"noise_pred{i}"].astype(np.float32))
        # x' = c3 (c1 x - c2 m) + c4 (c1 m + c2 x)  ->  the guided prediction m from the update the native path made
        c1, c2, c3, c4 = fused_step_coefficients(pipe.scheduler, t, n_inf)
        noise = (x_out - (c3 * c1 + c4 * c2) * x_in) / (c4 * c1 - c3 * c2)
        e_lat, e_noise = nrmse(x_out, want_lat), nrmse(noise, want_noise)
        per_frame = (noise - want_noise).pow(2).sum(dim=(0, 1, 3, 4)).sqrt() / want_noise.pow(2).sum(dim=(0, 1, 3, 4)).sqrt()
        print(f"[{case}] step {i} t={t}: latents nrmse {e_lat:.4e}  guided noise prediction nrmse {e_noise:.4e}  "
              f"worst frame {float(per_frame.max()):.4e}")
        assert e_noise < TOL_NOISE and e_lat < TOL_LAT and float(per_frame.max()) < 1.5 * TOL_NOISE, (case, i, e_lat, e_noise)
        x_in = x_out  # the next step starts from the native path's own latents, as in a real run


# The WHOLE 30-step trajectory at config #3's size, from one seeded start (tests/golden/steps_traj30.npz: the reference's own
# modules driven by the statements of its loop body, 2.5 h of host time once -- oracle/gen_fullsize_steps_golden.py traj30).
# Unlike the per-step cases above the native path is NOT re-synchronised to the reference: every step starts from its own
# latents, so the bound is a drift bound -- bf16 storage noise of 1.5e-2 per guided prediction, fed back 30 times through a
# random-weight network.  TRAJ_BOUNDS: step -> bound on the latents' nrmse, 2 x what MI355X measured when the fixture was made
# (profiles/r06_traj30.txt: 3.3e-4 after step 0, 2.6e-3 after 10, 6.0e-3 after 20, 7.3e-3 at the end -- the drift grows more
# slowly than the per-step noise would if it accumulated coherently); the latents must also keep the reference's scale (the
# reference's rms grows 1.00 -> 2.57 over the trajectory; within 1 %: measured 0.05 %).
TRAJ_BOUNDS = {0: 1e-3, 4: 2.5e-3, 9: 5e-3, 14: 9e-3, 19: 1.2e-2, 24: 1.4e-2, 29: 1.5e-2}


def test_thirty_step_trajectory_matches_the_reference_at_full_size(pipe_and_chan):
    path = os.path.join(GOLD, "steps_traj30.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/steps_traj30.npz not generated (oracle/gen_fullsize_steps_golden.py traj30)")
    pipe, chan = pipe_and_chan
    z = np.load(path)
    n_inf = int(z["num_inference_steps"])
    lat, pose, pl, clip, banks = FC.make_step_inputs("traj30", list(chan), lambda p: chan[p])
    eng = pipe.denoising_unet.engine()
    eng.set_reference_banks({k: v.cuda() for k, v in banks.items()}, do_cfg=True)
    eng._banks_from_modules = lambda: None
    got = []  # (the callback's first argument is the reference's rebound window index, not the step: see Pose2VideoPipeline.denoise)
    pipe.denoise(lat.clone().cuda(), pose.cuda(), pl.cuda(), clip.cuda(), n_inf, FC.GUIDANCE,
                 callback=lambda i, t, x: got.append((int(t), x.detach().float().cpu().clone())))
    torch.cuda.synchronize()
    assert len(got) == n_inf
    for i in sorted(TRAJ_BOUNDS):
        t, x = got[i]
        want = torch.from_numpy(z[f"latents{i}"].astype(np.float32))
        assert t == int(z[f"t{i}"]) and torch.isfinite(x).all()
        e = nrmse(x, want)
        rms, rms_ref = float(x.pow(2).mean().sqrt()), float(z[f"latent_rms{i}"])
        print(f"[traj30] step {i} t={t}: latents nrmse {e:.4e}  rms {rms:.4f} (reference {rms_ref:.4f})")
        assert e < TRAJ_BOUNDS[i] and abs(rms / rms_ref - 1) < 1e-2, (i, e, rms, rms_ref)
