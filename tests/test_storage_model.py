"""oracle/storage_model.py (the fp32 oracle with the native path's bf16 storage points) is pinned to the reference-pinned
oracle: with the rounding switched off it IS oracle_torch.unet3d_forward (up to fp32 re-association of the folded
LayerNorms); with it on, its distance from the fp32 result is the storage-format floor that tools/error_attribution.py
measured (about 1e-2), not more.  CPU only, tiny widths."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import oracle_torch as O  # noqa: E402
import storage_model as SM  # noqa: E402


def _case(seed=7, f=3, hh=8, ww=8):
    cfg = O.tiny_unet3d_cfg()
    sd = O.make_unet3d_weights(cfg, seed=0)
    g = torch.Generator().manual_seed(seed)
    sample = torch.randn(1, 4, f, hh, ww, generator=g).repeat(2, 1, 1, 1, 1)
    ehs = torch.cat([torch.zeros(1, 1, 768), torch.randn(1, 1, 768, generator=g)])
    pose = (torch.randn(1, 320, f, hh, ww, generator=g) * 0.5).repeat(2, 1, 1, 1, 1)
    banks = {}
    for p in O.transformer_locations(cfg):
        c = sd[p + ".norm.weight"].numel()
        banks[p] = torch.randn(2, hh * ww if c == 320 else (hh // 2) * (ww // 2), c, generator=g).half().float()
    return cfg, sd, (sample, 601, ehs, pose, banks)


def test_identity_rounding_reproduces_the_pinned_oracle():
    torch.set_grad_enabled(False)
    cfg, sd, (sample, t, ehs, pose, banks) = _case()
    taps_ref, taps = {}, {}
    ref = O.unet3d_forward(sd, cfg, sample, t, ehs, pose, banks, do_cfg=True, taps=taps_ref)
    got = SM.storage_model_forward(sd, cfg, sample, t, ehs, pose, banks, do_cfg=True, taps=taps, q=lambda x: x, gelu=F.gelu)
    assert set(taps) == set(taps_ref)
    for k in taps_ref:
        e = float((taps[k] - taps_ref[k]).norm() / taps_ref[k].norm())
        assert e < 2e-5, (k, e)
    assert float((got - ref).norm() / ref.norm()) < 2e-5
    # without a bank and without guidance (the other branches of the spatial transformer)
    ref = O.unet3d_forward(sd, cfg, sample, t, ehs, pose, None, do_cfg=False)
    got = SM.storage_model_forward(sd, cfg, sample, t, ehs, pose, None, do_cfg=False, q=lambda x: x, gelu=F.gelu)
    assert float((got - ref).norm() / ref.norm()) < 2e-5


def test_storage_model_sits_at_the_bf16_floor():
    torch.set_grad_enabled(False)
    cfg, sd, (sample, t, ehs, pose, banks) = _case(seed=8)
    ref = O.unet3d_forward(sd, cfg, sample, t, ehs, pose, banks, do_cfg=True)
    got = SM.storage_model_forward(sd, cfg, sample, t, ehs, pose, banks, do_cfg=True)
    e = float((got - ref).norm() / ref.norm())
    assert 2e-3 < e < 2e-2, e
    # the kernels' GELU form is within 5e-4 of the exact one wherever |gelu| matters
    x = torch.linspace(-8, 8, 4001)
    assert float((SM.native_gelu(x) - F.gelu(x)).abs().max()) < 1e-6 + 5e-4 * float(F.gelu(x).abs().max())
    assert float((SM.native_gelu(x) - F.gelu(x)).abs().max()) < 1e-5


def test_spatial_softmax_reference_rules():
    """the kernels' reference-maximum rules inside the storage model's attention (first-tile maximum; raised per 16-query
    group when a probability exceeds 2^8 at head dims 80 / 160; kept at head dim 40) are re-parameterisations of the same
    softmax: against the exact one they may only differ by the bf16 rounding of the probabilities and of the re-scaled
    query -- also when later tiles tower over the first (the raise branch)"""
    torch.set_grad_enabled(False)
    g = torch.Generator().manual_seed(3)
    m = SM.StorageModel({}, O.tiny_unet3d_cfg())
    for d in (40, 80, 160):
        n, H, L, Lk = 2, 2, 64, 320
        q = SM.bf16_round(torch.randn(n, H, L, d, generator=g))
        k = SM.bf16_round(torch.randn(n, H, Lk, d, generator=g))
        v = SM.bf16_round(torch.randn(n, H, Lk, d, generator=g))
        k[:, :, 200] = SM.bf16_round(q[:, :, 5] * 3.0)      # one key far above the first tile for query 5 (and its 16-group)
        k[:, :, 290] = SM.bf16_round(q[:, :, 40] * 4.0)     # another one, later, for another group
        ref = torch.nn.functional.scaled_dot_product_attention(q, k, v)
        got = m.sdpa(q, k, v, spatial=True)
        e = float((got - ref).norm() / ref.norm())
        assert e < 8e-3, (d, e)
        assert float((got[:, :, 5] - ref[:, :, 5]).norm() / ref[:, :, 5].norm()) < 2e-2
