#!/usr/bin/env python3
"""Same-box A/B of the two spatial-attention kernels at the config-#3 shapes (48 images, CFG layout: the second half
attends own + bank keys).  Prints ms and algorithmic TFLOP/s per kernel and shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from humanvid_amd import lib as hvlib
from humanvid_amd import ops

L, st, dev = hvlib.load(), hvlib.current_stream(), torch.device("cuda")
BF = torch.bfloat16


def timeit(fn, iters=3):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


n = 48
for D, N in [(40, 6144), (80, 1536), (160, 384), (160, 96)]:
    C, M = 8 * D, n * N
    qkv = torch.randn(M, 3 * C, device=dev).to(BF)
    kv2 = torch.randn(2 * N, 2 * C, device=dev).to(BF)
    o = torch.empty(M, C, dtype=BF, device=dev)
    sel = torch.tensor([-1] * (n // 2) + [1] * (n // 2), dtype=torch.int32, device=dev)
    flops = 4.0 * C * N * (N * n + N * n / 2)
    ms2 = timeit(lambda: ops.attention(L, st, qkv, qkv[:, C:], qkv[:, 2 * C:], o, n_images=n, heads=8, D=D, Lq=N, L1=N,
                                       ldq=3 * C, ldk=3 * C, ldvt=3 * C, ldo=C, k2=kv2, vt2=kv2[:, C:], ldk2=2 * C,
                                       ldvt2=2 * C, L2=N, bank_sel=sel, v_row_major=True))
    o2 = o.clone()
    qk = qkv[:, :2 * C].contiguous()
    vt = qkv[:, 2 * C:].t().contiguous()
    k2 = kv2[:, :C].contiguous()
    vt2 = kv2[:, C:].t().contiguous()
    ms1 = timeit(lambda: ops.attention(L, st, qk, qk[:, C:], vt, o, n_images=n, heads=8, D=D, Lq=N, L1=N, ldq=2 * C,
                                       ldk=2 * C, ldvt=M, ldo=C, k2=k2, vt2=vt2, ldk2=C, ldvt2=2 * N, L2=N, bank_sel=sel))
    diff = float((o.float() - o2.float()).abs().max())
    print(f"D={D:3d} N={N:5d}: round-1 kernel {ms1:8.3f} ms {flops / ms1 / 1e9:7.1f} TF/s | round-2 kernel {ms2:8.3f} ms "
          f"{flops / ms2 / 1e9:7.1f} TF/s | max|o1-o2| {diff:.4f}", flush=True)
