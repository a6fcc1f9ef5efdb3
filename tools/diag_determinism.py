#!/usr/bin/env python3
"""Run-to-run determinism check: temporal attention kernel alone, then the full-size UNet forward with the MFMA and the
VALU temporal kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from humanvid_amd import lib as hvlib
from humanvid_amd import ops

L = hvlib.load()
st = hvlib.current_stream()
dev = torch.device("cuda")
for D, P in [(40, 6144), (80, 1536), (160, 384)]:
    C = 8 * D
    M = 2 * 24 * P
    qkv = torch.randn(M, 3 * C, device=dev).to(torch.bfloat16)
    outs = []
    for mode in (1, 1, 0, 0):
        L.call("hv_set_tuning", 7, mode)
        o = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
        ops.temporal_attention(L, st, qkv, o, B=2, F=24, P=P, heads=8, D=D)
        torch.cuda.synchronize()
        outs.append(o.clone())
    print(f"D={D}: mfma run-to-run equal {torch.equal(outs[0], outs[1])}, valu equal {torch.equal(outs[2], outs[3])}, "
          f"mfma vs valu max diff {float((outs[0].float() - outs[2].float()).abs().max()):.4f}")
L.call("hv_set_tuning", 7, 1)

import bench
from humanvid_amd.unet3d import transformer_locations

unet, pg, cam = bench.build_models(dev)
eng = unet.engine()
g = torch.Generator(device=dev).manual_seed(5)
banks = {}
for loc in transformer_locations(unet):
    Cc = eng.w[loc + ".proj_in.w"].shape[0]
    lvl = {320: 0, 640: 1, 1280: 2}[Cc] if loc != "mid_block.attentions.0" else 3
    banks[loc] = torch.randn(2, (96 >> lvl) * (64 >> lvl), Cc, device=dev, generator=g).half().float()
eng.set_reference_banks(banks, do_cfg=True)
eng._banks_from_modules = lambda: None
gg = torch.Generator().manual_seed(42)
sample = torch.randn(1, 4, 24, 96, 64, generator=gg).repeat(2, 1, 1, 1, 1).cuda()
ehs = torch.cat([torch.zeros(1, 1, 768), torch.randn(1, 1, 768, generator=gg)]).cuda()
pose = (torch.randn(1, 320, 24, 96, 64, generator=gg) * 0.5).repeat(2, 1, 1, 1, 1).cuda()
for mode in (1, 0):
    L.call("hv_set_tuning", 7, mode)
    outs = []
    for i in range(3):
        o = unet(sample, 499, ehs, pose_cond_fea=pose, return_dict=False)[0]
        torch.cuda.synchronize()
        outs.append(o.clone())
    print(f"UNet forward, temporal mfma={mode}: run0==run1 {torch.equal(outs[0], outs[1])}, run1==run2 {torch.equal(outs[1], outs[2])}, "
          f"max diff {float((outs[0].float() - outs[1].float()).abs().max()):.4f} / {float((outs[1].float() - outs[2].float()).abs().max()):.4f}")
