"""CPU-side checks (no GPU): oracle vs the committed golden vectors, host logic of the native
package vs the oracle / golden vectors, the C-ABI surface, and the "no fallback" guards."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import oracle_torch as O  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")


# ------------------------------------------------------------------ oracle pinned to the reference
@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree not mounted")
def test_oracle_matches_reference_source():
    """Runs the reference's own model code (on the diffusers shim) against the oracle."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "oracle", "gen_golden.py"), "--check"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all checks passed" in r.stdout


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree not mounted")
def test_oracle_reference_net_matches_reference_source():
    """The reference's own UNet2DConditionModel (write mode) against oracle_torch.reference_net_banks."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "oracle", "gen_refnet_golden.py"), "--check"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ok oracle reference_net_banks == reference write pass" in r.stdout


def test_oracle_reproduces_golden_reference_net_banks():
    """tests/golden/refnet_sd15.npz holds the banks the REFERENCE's ReferenceNet wrote (oracle/gen_refnet_golden.py)."""
    z = np.load(os.path.join(GOLD, "refnet_sd15.npz"))
    cfg = dict(O.SD15_UNET3D_CFG)
    sd = O.make_reference_net_weights(cfg, seed=5)
    lat, clip = torch.from_numpy(z["lat"]), torch.from_numpy(z["clip"])
    banks = O.reference_net_banks(sd, cfg, lat, clip)  # the conditional entry alone
    names = [k[5:] for k in z.files if k.startswith("bank:")]
    assert len(names) == 16 and set(names) == set(banks)
    for n in names:
        want = torch.from_numpy(z["bank:" + n].astype(np.float32))[1:]
        assert float((banks[n] - want).abs().max()) <= 2e-3 * max(1.0, float(want.abs().max())), n  # fp16 fixture


def test_oracle_reproduces_golden_unet():
    z = np.load(os.path.join(GOLD, "unet3d_tiny.npz"))
    cfg = O.tiny_unet3d_cfg()
    sd = O.make_unet3d_weights(cfg, seed=0)
    sample = torch.from_numpy(z["sample"]).repeat(2, 1, 1, 1, 1)
    ehs = torch.cat([torch.zeros(1, 1, 768), torch.from_numpy(z["ehs"])])
    pose = torch.from_numpy(z["pose"]).repeat(2, 1, 1, 1, 1)
    banks = {k[5:]: torch.from_numpy(z[k].astype(np.float32)) for k in z.files if k.startswith("bank:")}
    out = O.unet3d_forward(sd, cfg, sample, int(z["t"]), ehs, pose, banks, do_cfg=True)
    assert float((out - torch.from_numpy(z["out"])).abs().max()) < 5e-5


def test_oracle_reproduces_golden_conditioning():
    z = np.load(os.path.join(GOLD, "pose_guider.npz"))
    out = O.pose_guider_forward(O.make_pose_guider_weights(), torch.from_numpy(z["cond"]))
    assert float((out - torch.from_numpy(z["out"])).abs().max()) < 1e-5
    z = np.load(os.path.join(GOLD, "camera_encoder.npz"))
    out = O.camera_encoder_forward(O.make_camera_encoder_weights(), torch.from_numpy(z["plucker"]))
    assert float((out - torch.from_numpy(z["out"])).abs().max()) < 1e-4


# ------------------------------------------------------------------ native host logic
def test_state_dict_grammar_matches_reference_manifest():
    from humanvid_amd.arch import SD15_INFERENCE_V2
    from humanvid_amd.unet3d import UNet3DConditionModel, transformer_locations

    with torch.device("meta"):
        m = UNet3DConditionModel(**SD15_INFERENCE_V2)
    mine = {k: list(v.shape) for k, v in m.state_dict().items()}
    ref = json.load(open(os.path.join(GOLD, "unet3d_sd15_keys.json")))
    assert mine == ref and len(ref) == 1274
    full = dict(O.SD15_UNET3D_CFG)
    assert transformer_locations(m) == O.transformer_locations(full)
    assert m.in_channels == 4 and m.config.cross_attention_dim == 768  # config attribute fallback


def test_containers_refuse_eager_forward_and_cpu_tensors():
    from humanvid_amd.unet3d import UNet3DConditionModel

    cfg = O.tiny_unet3d_cfg()
    net = UNet3DConditionModel(**dict(cfg, use_inflated_groupnorm=True, motion_module_type="Vanilla",
                                      unet_use_cross_frame_attention=False, unet_use_temporal_attention=False))
    with pytest.raises(RuntimeError):
        net.conv_in(torch.zeros(1, 4, 1, 8, 8))
    with pytest.raises(RuntimeError):  # no GPU here / CPU tensors: the product path must fail loudly
        net(torch.zeros(2, 4, 2, 8, 8), 10, torch.zeros(2, 1, 768))


def test_product_loader_only_knows_the_hip_library():
    from humanvid_amd import lib

    assert lib.LIB_PATH.endswith(os.path.join("humanvid_amd", "lib", "libhumanvid_hip.so"))
    src = open(os.path.join(REPO, "humanvid_amd", "lib.py")).read()
    assert "emu" not in src and "oracle" not in src
    for root, _, files in os.walk(os.path.join(REPO, "humanvid_amd")):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(root, f)).read()
                assert "oracle_torch" not in text and "import oracle" not in text, f"{f} imports the oracle"


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, "include", "humanvid_hip.h")).read()
    declared = sorted(set(re.findall(r"^\s*(?:const char\*|int)\s+(hv_[a-z0-9_]+)\s*\(", hdr, flags=re.M)))
    assert len(declared) >= 20
    so = os.path.join(REPO, "humanvid_amd", "lib", "libhumanvid_hip.so")
    if not os.path.exists(so):
        sys.path.insert(0, REPO)
        import __graft_entry__

        __graft_entry__.build()
    dll = ctypes.CDLL(so)
    for name in declared:
        assert hasattr(dll, name), f"{name} is declared in include/humanvid_hip.h but not exported"
    from humanvid_amd import _abi

    assert set(_abi.PROTOTYPES) == set(declared)
    _abi.HvLibrary(so)  # struct-size mirror check


def test_ddim_and_windows_match_oracle_and_golden():
    from humanvid_amd.scheduler import DDIMScheduler, uniform

    s = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=False, steps_offset=1,
                      prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")
    o = O.DDIM()
    for n in (4, 25, 30):
        s.set_timesteps(n)
        o.set_timesteps(n)
        assert s.timesteps.tolist() == o.timesteps.tolist()
        x, v = torch.randn(1, 4, 3, 8, 8), torch.randn(1, 4, 3, 8, 8)
        for t in s.timesteps.tolist()[:3] + s.timesteps.tolist()[-1:]:
            assert torch.allclose(s.step(v, t, x).prev_sample, o.step(v, t, x), atol=1e-6)
    rep = json.load(open(os.path.join(GOLD, "oracle_pin_report.json")))
    s.set_timesteps(30)
    assert s.timesteps.tolist() == rep["ddim_timesteps_30"]
    win = json.load(open(os.path.join(GOLD, "context_windows.json")))
    for key, want in win.items():
        nf, ov = (int(x) for x in key.split(":"))
        assert list(uniform(0, 30, nf, 24, 1, ov)) == want
    assert list(uniform(0, 30, 48, 24, 1, 4)) == [list(range(24)), list(range(20, 44)),
                                                  list(range(40, 48)) + list(range(16))]


def test_fused_step_coefficients_cover_v_and_epsilon_prediction():
    """hv_cfg_ddim_step evaluates x0 = c1 x - c2 m, eps = c1 m + c2 x, x' = c3 x0 + c4 eps; the coefficient sets for
    v-prediction (inference_v2.yaml) and epsilon-prediction (inference_v1.yaml:18-23, diffusers' default) must reproduce
    the published DDIM update (eta = 0), from this module's scheduler and from a diffusers-style object (attributes under
    `.config`, no step_coefficients)."""
    import types

    from humanvid_amd.scheduler import DDIMScheduler, fused_step_coefficients

    def fused(c, x, m):
        c1, c2, c3, c4 = c
        return c3 * (c1 * x - c2 * m) + c4 * (c1 * m + c2 * x)

    x, m = torch.randn(1, 4, 3, 8, 8, dtype=torch.float64), torch.randn(1, 4, 3, 8, 8, dtype=torch.float64)
    for kw in (dict(prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing"),
               dict(prediction_type="epsilon", timestep_spacing="leading")):  # inference_v2 / inference_v1
        s = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=False, steps_offset=1, **kw)
        s.set_timesteps(25)
        like = types.SimpleNamespace(alphas_cumprod=s.alphas_cumprod, final_alpha_cumprod=s.final_alpha_cumprod,
                                     config=dict(prediction_type=kw["prediction_type"], num_train_timesteps=1000,
                                                 clip_sample=False),
                                     step=lambda model_output, timestep, sample, eta=0.0: None)  # diffusers' DDIM shape
        for t in s.timesteps.tolist()[:3] + s.timesteps.tolist()[-2:]:
            c = fused_step_coefficients(s, t, 25)
            assert c == fused_step_coefficients(like, t, 25)
            want = s.step(m, t, x).prev_sample  # tensor form of the published update
            assert torch.allclose(fused(c, x, m), want, atol=1e-5, rtol=1e-5)
    bad = DDIMScheduler(beta_start=0.00085, beta_end=0.012, clip_sample=False, prediction_type="epsilon",
                        rescale_betas_zero_snr=True, timestep_spacing="trailing")
    bad.set_timesteps(25)
    with pytest.raises(ValueError):
        fused_step_coefficients(bad, 999, 25)  # zero terminal SNR has no epsilon form
    with pytest.raises(NotImplementedError):
        fused_step_coefficients(types.SimpleNamespace(config=dict(prediction_type="sample"), alphas_cumprod=s.alphas_cumprod), 10, 25)
    # diffusers-style schedulers that are NOT DDIM also carry alphas_cumprod: refused, not sampled with DDIM coefficients
    for extra in (dict(order=2), dict(sigmas=torch.ones(26)), dict(model_outputs=[None, None]), dict(ets=[])):
        other = types.SimpleNamespace(alphas_cumprod=s.alphas_cumprod, config=dict(prediction_type="epsilon"), **extra)
        with pytest.raises(NotImplementedError):
            fused_step_coefficients(other, 10, 25)
    dpm = types.SimpleNamespace(alphas_cumprod=s.alphas_cumprod, config=dict(prediction_type="epsilon", solver_order=2,
                                                                               algorithm_type="dpmsolver++"))
    with pytest.raises(NotImplementedError):
        fused_step_coefficients(dpm, 10, 25)

    # first-order schedulers WITHOUT a sigma table that are not DDIM (DDPM: ancestral; LCM: consistency): the check is positive
    class DDPMScheduler:  # diffusers' shape: alphas_cumprod, order 1, step() without eta, no final_alpha_cumprod
        order = 1

        def __init__(self):
            self.alphas_cumprod, self.config = s.alphas_cumprod, dict(prediction_type="epsilon")

        def step(self, model_output, timestep, sample, generator=None):
            raise AssertionError

    class LCMScheduler(DDPMScheduler):  # carries final_alpha_cumprod, still no eta
        def __init__(self):
            super().__init__()
            self.final_alpha_cumprod = torch.tensor(1.0)

    for cls in (DDPMScheduler, LCMScheduler):
        with pytest.raises(NotImplementedError):
            fused_step_coefficients(cls(), 10, 25)

    class MyScheduler(LCMScheduler):  # DDIM by shape (final_alpha_cumprod + step(..., eta)) under another name: accepted
        def step(self, model_output, timestep, sample, eta=0.0):
            raise AssertionError

    assert len(fused_step_coefficients(MyScheduler(), 10, 25)) == 4


def test_camera_front_end_matches_golden():
    from humanvid_amd.camera import Camera, cameras_to_embedding

    z = np.load(os.path.join(GOLD, "plucker.npz"))
    img_size = tuple(int(v) for v in z["img_size"])
    cams = [Camera(list(r), "test", img_size) for r in z["rows"]]
    got = cameras_to_embedding(cams, img_size)
    assert got.shape == z["out"].shape
    assert float((got - torch.from_numpy(z["out"])).abs().max()) < 1e-6
    with pytest.raises(AssertionError):
        Camera([0.0] * 8, "test", img_size)
    with pytest.raises(ValueError):
        Camera([0.0] * 7 + [1.0, 1.0, 1.0], "unknown_dataset", img_size)


def test_weight_packing_algebra():
    from humanvid_amd import packing

    g = torch.Generator().manual_seed(0)
    C, N, M = 64, 48, 10
    x = torch.randn(M, C, generator=g) + 0.3
    w, b = torch.randn(N, C, generator=g) / 8, torch.randn(N, generator=g)
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    wf, colsum, bf = packing.fold_layernorm(w, b, gamma, beta)
    mean, var = x.mean(1, keepdim=True), x.var(1, unbiased=False, keepdim=True)
    rstd = (var + 1e-5).rsqrt()
    got = rstd * (x @ wf.float().t() - mean * colsum[None]) + bf[None]
    want = torch.nn.functional.layer_norm(x, (C,), gamma, beta) @ w.t() + b
    assert float((got - want).abs().max()) < 2e-2  # folded weights are bf16
    order = packing.geglu_row_order(64)
    assert order[:16].tolist() == list(range(16)) and order[16:32].tolist() == list(range(32, 48))
    wc = torch.randn(8, 5, 3, 3, generator=g)
    pk = packing.pack_conv3x3(wc)
    assert pk.shape == (8, 9, 32) and float(pk[:, :, 5:].abs().max()) == 0
    assert torch.equal(pk[3, 4, :5].float(), wc[3, :, 1, 1].to(torch.bfloat16).float())


def test_flop_model_matches_survey():
    from humanvid_amd.arch import DEFAULT_UNET3D_CONFIG, SD15_INFERENCE_V2
    from humanvid_amd.workload import unet3d_flops

    cfg = dict(DEFAULT_UNET3D_CONFIG)
    cfg.update(SD15_INFERENCE_V2)
    assert abs(unet3d_flops(cfg, 2, 24, 96, 64)["total"] / 1e12 - 88.6) < 0.2
    assert abs(unet3d_flops(cfg, 2, 24, 96, 64, True)["total"] / 1e12 - 107.4) < 0.2
    assert abs(unet3d_flops(cfg, 2, 16, 64, 64)["total"] / 1e12 - 36.4) < 0.2


def test_reference_control_pairs_banks_like_the_reference():
    from humanvid_amd.reference_control import ReferenceAttentionControl
    from humanvid_amd.unet2d import BasicTransformerBlock, UNet2DConditionModel
    from humanvid_amd.unet3d import TemporalBasicTransformerBlock, UNet3DConditionModel

    cfg = O.tiny_unet3d_cfg()
    with torch.device("meta"):
        den = UNet3DConditionModel(**dict(cfg, use_inflated_groupnorm=True, motion_module_type="Vanilla",
                                          unet_use_cross_frame_attention=False, unet_use_temporal_attention=False))
        ref = UNet2DConditionModel(block_out_channels=(320, 640), layers_per_block=1, cross_attention_dim=768,
                                   down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                                   up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"))
    assert set(ref.state_dict()) == set(O.make_reference_net_weights(cfg))
    writer = ReferenceAttentionControl(ref, mode="write", do_classifier_free_guidance=True, fusion_blocks="full")
    reader = ReferenceAttentionControl(den, mode="read", do_classifier_free_guidance=True, fusion_blocks="full")
    wb = [m for m in ref.modules() if isinstance(m, BasicTransformerBlock)]
    for i, m in enumerate(wb):
        m.bank.append(torch.full((2, 4, m.norm1.normalized_shape[0]), float(i)))
    reader.update(writer)
    names = {id(m): n for n, m in den.named_modules()}
    wnames = {id(m): n for n, m in ref.named_modules()}
    for r_, w_ in zip(reader._modules(den, kind=TemporalBasicTransformerBlock), writer._modules(ref, kind=BasicTransformerBlock)):
        assert names[id(r_)] == wnames[id(w_)]  # identical location => identical pairing to the reference's sort
        assert r_.bank[0].dtype == torch.float16 and torch.equal(r_.bank[0].float(), w_.bank[0])
    assert den._reference_mode == dict(mode="read", do_cfg=True, fusion_blocks="full")
    reader.clear()
    assert all(len(m.bank) == 0 for m in den.modules() if isinstance(m, TemporalBasicTransformerBlock))
    with pytest.raises(AssertionError):
        ReferenceAttentionControl(den, mode="bogus")


def test_camera_params_are_what_ray_condition_consumes():
    """cameras_to_params (the input of the on-device Pluecker front-end) and cameras_to_embedding (the reference's map,
    scripts/pose2vid.py:45-83) describe the same cameras: ray_condition(K, c2w) == embedding."""
    from humanvid_amd import camera

    rows = [[0.0] * 7 + [1.0, 1.788079, 1.0, 1.0] for _ in range(4)]
    for i, r in enumerate(rows):  # TUM: tx ty tz qx qy qz qw + intrinsics; move the camera a little per frame
        r[0], r[1], r[2] = 0.1 * i, -0.05 * i, 0.02 * i
    cams = [camera.Camera(r, "test", (48, 32)) for r in rows]
    K, c2w = camera.cameras_to_params(cams, (48, 32))
    emb = camera.cameras_to_embedding(cams, (48, 32))
    assert K.shape == (3, 4) and c2w.shape == (3, 4, 4) and emb.shape == (1, 3, 6, 32, 48)
    pl = camera.ray_condition(K[None], c2w[None], 32, 48, device="cpu")[0].permute(0, 3, 1, 2)
    assert torch.equal(pl, emb[0])


def test_library_override_env_is_honoured(monkeypatch, tmp_path):
    """HUMANVID_HIP_LIB points the loader at another build of the same C ABI (same-run A/Bs); a wrong path fails loudly"""
    import humanvid_amd.lib as hvlib

    monkeypatch.setattr(hvlib, "_LIB", None)
    monkeypatch.setenv("HUMANVID_HIP_LIB", str(tmp_path / "does_not_exist.so"))
    with pytest.raises(OSError):
        hvlib.load()
    monkeypatch.setattr(hvlib, "_LIB", None)


def test_loop_body_goldens_counters_and_callback_contract():
    """tests/golden/steps_*.npz (the reference's own loop body, oracle/gen_fullsize_steps_golden.py): the per-frame window
    counters the reference accumulated equal the coverage of this package's window table, and the golden timesteps are the
    ones this package's scheduler walks (t = 999 first, t = 32 last of 30)."""
    import sys

    sys.path.insert(0, os.path.dirname(__file__))
    import fullsize_case as FC
    from humanvid_amd.scheduler import DDIMScheduler, get_context_scheduler

    s = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=False, steps_offset=1,
                      prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")
    for case, geo in FC.STEP_CASES.items():
        path = os.path.join(GOLD, f"steps_{case}.npz")
        if "keep" in geo and not os.path.exists(path):
            continue  # the 30-step trajectory fixture is optional (2.5 h of host time to make)
        z = np.load(path)
        s.set_timesteps(int(z["num_inference_steps"]))
        ts = [int(t) for t in s.timesteps.tolist()]
        windows = list(get_context_scheduler("uniform")(0, int(z["num_inference_steps"]), geo["F"], 24, 1, 4))
        cover = np.zeros(geo["F"])
        for c in windows:
            cover[list(c)] += 1
        for i in geo["steps"]:
            assert int(z[f"t{i}"]) == ts[i]
            if f"counter{i}" in z:  # (the trajectory fixture keeps latents and timesteps only)
                assert np.array_equal(z[f"counter{i}"], cover), (case, z[f"counter{i}"], cover)
    assert ts[0] == 999 and ts[-1] == 32
    assert len(windows) == 1  # (the last case is a single 24-frame window)


def test_interleaved_replay_order_and_streams(monkeypatch):
    """StepRecorder.replay_interleaved (exchange / compute overlap of the two CFG halves): item i of
    every half is issued before item i + 1 of any, each half on its own stream -- so the collectives reach the communicator
    in the order A1 B1 A2 B2 ... on every rank (the property that keeps the ranks' collective sequences identical) -- and a
    recording whose item kinds differ from the other half's is refused."""
    import contextlib

    from humanvid_amd.pipeline import StepRecorder

    log, cur = [], [None]

    class FakeStream:
        def __init__(self, name):
            self.name, self.cuda_stream = name, name

    class FakeLib:
        def call(self, fn, handle, stream):
            assert fn == "hv_cmdlist_run" and stream == cur[0].name  # launched on the stream it is issued under
            log.append(("k", handle, stream))

    @contextlib.contextmanager
    def fake_stream_ctx(s):
        prev, cur[0] = cur[0], s
        try:
            yield
        finally:
            cur[0] = prev

    monkeypatch.setattr(torch.cuda, "stream", fake_stream_ctx)
    recs = []
    for half in ("A", "B"):
        r = StepRecorder(FakeLib())
        r.items = [("k", half + "0"), ("c", lambda h=half: log.append(("c", h + "1", cur[0].name))), ("k", half + "2"),
                   ("c", lambda h=half: log.append(("c", h + "3", cur[0].name)))]
        recs.append(r)
    StepRecorder.replay_interleaved(recs, [FakeStream("sA"), FakeStream("sB")])
    assert log == [("k", "A0", "sA"), ("k", "B0", "sB"), ("c", "A1", "sA"), ("c", "B1", "sB"),
                   ("k", "A2", "sA"), ("k", "B2", "sB"), ("c", "A3", "sA"), ("c", "B3", "sB")]
    recs[1].items = recs[1].items[:3] + [("k", "B3")]
    with pytest.raises(AssertionError):
        StepRecorder.replay_interleaved(recs, [FakeStream("sA"), FakeStream("sB")])
