#!/usr/bin/env python3
"""Full-size check of the CFG-parallel axis' building block: the B = 1 forward of each CFG half (Engine.clone_for_half) against
the matching half of the B = 2 forward on the same inputs -- bit for bit, finite -- at config #3's size with the bench's
synthetic weights and banks.  Also times hv_attention40 per launch in the three forms (python tools/half_forward_check.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from humanvid_amd import lib as hvlib
from humanvid_amd import ops
from humanvid_amd.unet3d import transformer_locations

dev = hvlib.require_gpu()
L = hvlib.load()
unet, pg, cam = bench.build_models(dev)
F, h, w = 24, 96, 64
eng = unet.engine()
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
g = torch.Generator().manual_seed(42)
lat = torch.randn(1, 4, F, h, w, generator=g).to(dev)
cond = (torch.randn(F, h, w, 320, generator=g) * 0.5).to(dev).to(torch.bfloat16)
st = hvlib.current_stream()
x2 = torch.zeros(2 * F, h, w, 32, dtype=torch.bfloat16, device=dev)
ops.pack_ncfhw(L, st, lat, x2, rep=2, frames=torch.arange(F, dtype=torch.int32, device=dev))
t_dev = torch.full((2,), 601.0, device=dev)
y2 = eng.forward_nhwc(x2, t_dev, cond, B=2, F=F).clone()
torch.cuda.synchronize()
print("B = 2 forward finite:", bool(torch.isfinite(y2.float()).all()))
for hf in (0, 1):
    eh = eng.clone_for_half(hf)
    y1 = eh.forward_nhwc(x2[hf * F:(hf + 1) * F].contiguous(), t_dev[hf:hf + 1], cond, B=1, F=F).clone()
    torch.cuda.synchronize()
    ref = y2[hf * F:(hf + 1) * F]
    same = torch.equal(y1, ref)
    d = float((y1.float() - ref.float()).abs().max())
    print(f"half {hf}: finite {bool(torch.isfinite(y1.float()).all())}, equal to the B = 2 forward's half bit for bit: {same} (max abs diff {d:.3e})")

# ---- hv_attention40 on the step's REAL operands: capture the first head-dim-40 attention call of the conditional half's forward
# and of the B = 2 forward, then time each call standalone (and the careful-pass fallback: tuning 0 = 2 is the generic kernel)
from humanvid_amd import engine as E

captured = {}
real_attention = ops.attention


def spy(lib, stream, q, k, vt, o, **kw):
    if kw.get("D") == 40:
        captured[(kw["n_images"], len([c for c in captured if c[0] == kw["n_images"]]))] = (q.clone(), k, vt.clone(), torch.empty_like(o), dict(kw))
    return real_attention(lib, stream, q, k, vt, o, **kw)


E.ops.attention = spy
eng.forward_nhwc(x2, t_dev, cond, B=2, F=F)
eh = eng.clone_for_half(1)
eh.forward_nhwc(x2[F:].contiguous(), t_dev[1:2], cond, B=1, F=F)
torch.cuda.synchronize()
E.ops.attention = real_attention


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return ts


for (n, site), (q, k, vt, o, kw) in sorted(captured.items()):
    C = 320
    kk = q[:, C:]  # (the K operand is a column slice of the fused q|k buffer)
    for tune in (0, 2):
        L.call("hv_set_tuning", 0, tune)
        ts = timeit(lambda: real_attention(L, st, q, kk, vt, o, **kw))
        print(f"real operands, call site {site}, n = {n}, bank_sel = {kw['bank_sel'].tolist()[:2]}..{kw['bank_sel'].tolist()[-2:]}, kernel {'hv_attention40' if tune == 0 else 'generic'}: "
              + " ".join(f"{t:.3f}" for t in ts) + " ms", flush=True)
    L.call("hv_set_tuning", 0, 0)
