"""End-to-end parity of the native UNet3DConditionModel.forward on MI355X.

 * against the committed golden fixture tests/golden/unet3d_tiny.npz, whose output was produced by
   the REFERENCE's own UNet3DConditionModel + ReferenceAttentionControl (oracle/gen_golden.py),
 * against the oracle restatement on further seeded geometries (incl. head dim 160).

Tolerance (stated): the native path stores activations/weights in bf16 with fp32 accumulation,
the reference runs fp32 -> normalised RMSE of the UNet output <= 2e-2 (measured ~5e-3..1e-2),
no NaN/Inf.  Inputs are fixed-seed synthetic latents/weights (no checkpoints are reachable).
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import oracle_torch as O  # noqa: E402  (test infrastructure)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 2e-2


def nrmse(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm())


def build_native(cfg, sd):
    from humanvid_amd.unet3d import UNet3DConditionModel

    kw = dict(cfg)
    kw.update(use_inflated_groupnorm=True, unet_use_cross_frame_attention=False, unet_use_temporal_attention=False,
              motion_module_type="Vanilla")
    net = UNet3DConditionModel(**kw)
    missing, unexpected = net.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return net.to("cuda")


def run_native(net, sample, t, ehs, pose, banks, do_cfg=True, taps=None):
    eng = net.engine()
    if taps is not None:
        eng.tap = lambda name, x: taps.__setitem__(name, x.float().permute(0, 3, 1, 2).cpu())
    eng.set_reference_banks({k: v.cuda() for k, v in banks.items()} if banks else None, do_cfg=do_cfg)
    eng.bank_version = "pinned"  # banks were set explicitly
    net._reference_mode = None
    eng._banks_from_modules = lambda: None
    out = net(sample.cuda(), t, ehs.cuda(), pose_cond_fea=None if pose is None else pose.cuda(), return_dict=False)[0]
    torch.cuda.synchronize()
    eng.tap = None
    assert torch.isfinite(out).all()
    return out


def test_unet_matches_reference_golden():
    z = np.load(os.path.join(GOLD, "unet3d_tiny.npz"))
    cfg = O.tiny_unet3d_cfg()
    sd = O.make_unet3d_weights(cfg, seed=0)
    net = build_native(cfg, sd)
    sample = torch.from_numpy(z["sample"]).repeat(2, 1, 1, 1, 1)
    ehs = torch.cat([torch.zeros(1, 1, 768), torch.from_numpy(z["ehs"])])
    pose = torch.from_numpy(z["pose"]).repeat(2, 1, 1, 1, 1)
    banks = {k[5:]: torch.from_numpy(z[k].astype(np.float32)) for k in z.files if k.startswith("bank:")}
    taps = {}
    out = run_native(net, sample, int(z["t"]), ehs, pose, banks, taps=taps)
    e = nrmse(out, torch.from_numpy(z["out"]))
    print("unet3d tiny vs reference golden: nrmse", e)
    assert e < TOL, e
    # the intermediate activations the fixture holds (oracle taps, pinned to the reference at 5e-6): a wrong low-gain
    # branch cannot hide under the output norm
    names = [k[4:] for k in z.files if k.startswith("tap:")]
    assert len(names) >= 4
    for name in names:
        et = nrmse(taps[name], torch.from_numpy(z["tap:" + name].astype(np.float32)))
        print(f"  tap {name}: nrmse {et:.3e}")
        assert et < TOL, (name, et)


@pytest.mark.parametrize("geom", ["three_level_d160", "tiny_no_bank", "tiny_f8"])
def test_unet_matches_oracle(geom):
    g = torch.Generator().manual_seed(7)
    if geom == "three_level_d160":
        cfg = O.tiny_unet3d_cfg(down_block_types=("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
                                up_block_types=("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"),
                                block_out_channels=(320, 640, 1280), motion_module_resolutions=(1, 2, 4, 8))
        f, hh, ww = 3, 16, 16
    elif geom == "tiny_f8":
        cfg, f, hh, ww = O.tiny_unet3d_cfg(), 8, 12, 8
    else:
        cfg, f, hh, ww = O.tiny_unet3d_cfg(), 2, 8, 8
    sd = O.make_unet3d_weights(cfg, seed=11)
    net = build_native(cfg, sd)
    sample = torch.randn(1, 4, f, hh, ww, generator=g).repeat(2, 1, 1, 1, 1)
    ehs = torch.cat([torch.zeros(1, 1, 768), torch.randn(1, 1, 768, generator=g)])
    pose = (torch.randn(1, 320, f, hh, ww, generator=g) * 0.5).repeat(2, 1, 1, 1, 1)
    banks = {}
    if geom != "tiny_no_bank":
        for p in O.transformer_locations(cfg):
            c = sd[p + ".norm.weight"].numel()
            lvl = {320: 0, 640: 1, 1280: 2}[c]
            banks[p] = torch.randn(2, (hh >> lvl) * (ww >> lvl), c, generator=g).half().float()
    t = 333
    ref = O.unet3d_forward(sd, cfg, sample, t, ehs, pose, banks, do_cfg=True)
    out = run_native(net, sample, t, ehs, pose, banks)
    e = nrmse(out, ref)
    print(geom, "nrmse", e)
    assert e < TOL, e


def test_reference_net_banks_match_oracle():
    from humanvid_amd.reference_control import ReferenceAttentionControl
    from humanvid_amd.unet2d import UNet2DConditionModel

    cfg = O.tiny_unet3d_cfg()
    sd = O.make_reference_net_weights(cfg, seed=5)
    kw = dict(block_out_channels=cfg["block_out_channels"], layers_per_block=cfg["layers_per_block"],
              cross_attention_dim=768, attention_head_dim=8,
              down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"))
    net = UNet2DConditionModel(**kw)
    missing, unexpected = net.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    net = net.to("cuda")
    g = torch.Generator().manual_seed(3)
    lat = torch.randn(1, 4, 8, 8, generator=g).repeat(2, 1, 1, 1)
    ehs = torch.cat([torch.zeros(1, 1, 768), torch.randn(1, 1, 768, generator=g)])
    writer = ReferenceAttentionControl(net, do_classifier_free_guidance=True, mode="write", fusion_blocks="full")
    net(lat.cuda(), torch.zeros((), device="cuda"), encoder_hidden_states=ehs.cuda(), return_dict=False)
    torch.cuda.synchronize()
    ref = O.reference_net_banks(sd, cfg, lat, ehs)
    from humanvid_amd.unet2d import BasicTransformerBlock

    blocks = {n.rsplit(".transformer_blocks.0", 1)[0]: m for n, m in net.named_modules()
              if isinstance(m, BasicTransformerBlock)}
    assert set(blocks) == set(ref)
    for loc, want in ref.items():
        assert len(blocks[loc].bank) == 1
        e = nrmse(blocks[loc].bank[0], want)
        print(loc, "bank nrmse", e)
        assert e < TOL, (loc, e)
    writer.clear()
    assert all(len(m.bank) == 0 for m in blocks.values())
