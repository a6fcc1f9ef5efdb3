#!/usr/bin/env python3
"""In-context duration of the level-0 spatial attention launches of the conditional-half forward (B = 1), engine built
without / with a (one-rank) FrameShard -- bench.py --cfg-half 1 measured 5.6 ms per launch where the same call on the same
operands takes 2.45 ms standalone (tools/half_forward_check.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import bench
from humanvid_amd import engine as E
from humanvid_amd import lib as hvlib
from humanvid_amd import ops
from humanvid_amd.runner import FrameShard
from humanvid_amd.unet3d import transformer_locations

dev = hvlib.require_gpu()
L = hvlib.load()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", device_id=dev)
unet, pg, cam = bench.build_models(dev)
F, h, w = 24, 96, 64
g = torch.Generator().manual_seed(42)
lat = torch.randn(1, 4, F, h, w, generator=g).to(dev)
cond = (torch.randn(F, h, w, 320, generator=g) * 0.5).to(dev).to(torch.bfloat16)
st = hvlib.current_stream()
x2 = torch.zeros(2 * F, h, w, 32, dtype=torch.bfloat16, device=dev)
ops.pack_ncfhw(L, st, lat, x2, rep=2, frames=torch.arange(F, dtype=torch.int32, device=dev))
t_dev = torch.full((2,), 601.0, device=dev)
real_attention = ops.attention
events = []


def spy(lib, stream, q, k, vt, o, **kw):
    if kw.get("D") != 40:
        return real_attention(lib, stream, q, k, vt, o, **kw)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    r = real_attention(lib, stream, q, k, vt, o, **kw)
    b.record()
    events.append((a, b, kw["n_images"], q.data_ptr(), vt.data_ptr(), kw["k2"].data_ptr(), kw["ldq"], kw["ldvt"], kw["ldvt2"]))
    return r


E.ops.attention = spy
for label, shard in (("no shard", None), ("one-rank FrameShard", FrameShard())):
    eng = E.UNet3DEngine(unet, shard=shard)
    gb = torch.Generator(device=dev).manual_seed(5)
    banks = {}
    for loc in transformer_locations(unet):
        C = eng.w[loc + ".proj_in.w"].shape[0]
        lvl = {320: 0, 640: 1, 1280: 2}[C] if loc != "mid_block.attentions.0" else 3
        banks[loc] = torch.randn(2, (h >> lvl) * (w >> lvl), C, device=dev, generator=gb).half().float()
    eng.set_reference_banks(banks, do_cfg=True)
    eng._banks_from_modules = lambda: None
    e = torch.randn(1, 1, 768, generator=torch.Generator().manual_seed(2)).to(dev)
    eng.set_encoder_hidden_states(torch.cat([torch.zeros_like(e), e], dim=0))
    for hf in (1, 0):
        eh = eng.clone_for_half(hf)
        for rep in range(2):
            events.clear()
            eh.forward_nhwc(x2[hf * F:(hf + 1) * F].contiguous(), t_dev[hf:hf + 1], cond, B=1, F=F)
            torch.cuda.synchronize()
        print(f"{label}, half {hf}: " + " ".join(f"{a.elapsed_time(b):.3f}" for a, b, *_ in events) + " ms;  q/vt/k2 addresses mod 2 MiB: "
              + " ".join(f"{(ev[3] % (1 << 21)) >> 10}K/{(ev[4] % (1 << 21)) >> 10}K/{(ev[5] % (1 << 21)) >> 10}K" for ev in events[:2])
              + f"  ldq {events[0][6]} ldvt {events[0][7]} ldvt2 {events[0][8]}", flush=True)
