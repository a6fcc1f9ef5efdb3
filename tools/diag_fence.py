#!/usr/bin/env python3
"""Root-cause harness for the temporal-attention MFMA fault of round 1 (HV_TEMPORAL_FENCE / HV_TEMPORAL_TAIL in
humanvid_amd/csrc/hv_temporal.h).  For every diagnosis build tools/bin/lib_f<fence>t<tail>.so (tools/build_fence_variants.sh)
it runs, in a fresh process, the temporal kernel at the three config-#3 shapes REPS times on the same input and reports
run-to-run bit equality and the distance to the VALU kernel (which has no MFMA).

    python tools/diag_fence.py            # driver: one subprocess per variant, prints a table
"""
import glob
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPS = 8


def child():
    sys.path.insert(0, REPO)
    import torch

    from humanvid_amd import lib as hvlib
    from humanvid_amd import ops

    L = hvlib.load()
    st = hvlib.current_stream()
    dev = torch.device("cuda")
    res = {}
    for D, P in [(40, 6144), (80, 1536), (160, 384)]:
        C, M = 8 * D, 2 * 24 * P
        qkv = torch.randn(M, 3 * C, device=dev, generator=torch.Generator(device=dev).manual_seed(D)).to(torch.bfloat16)
        L.call("hv_set_tuning", 7, 0)
        ref = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
        ops.temporal_attention(L, st, qkv, ref, B=2, F=24, P=P, heads=8, D=D)
        L.call("hv_set_tuning", 7, 1)
        outs = []
        for _ in range(REPS):
            o = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
            ops.temporal_attention(L, st, qkv, o, B=2, F=24, P=P, heads=8, D=D)
            outs.append(o)
        torch.cuda.synchronize()
        neq = sum(int(not torch.equal(outs[0], o)) for o in outs[1:])
        bad_rows = int(((outs[0].float() - ref.float()).abs().amax(dim=1) > 0.05).sum())
        worst = max(float((o.float() - ref.float()).abs().max()) for o in outs)
        res[str(D)] = dict(runs_differing_from_run0=neq, rows_off_vs_valu=bad_rows, max_abs_vs_valu=round(worst, 4))
    print("RESULT " + json.dumps(res))


def main():
    libs = sorted(glob.glob(os.path.join(REPO, "tools", "bin", "lib_f*t*.so")))
    print(f"{'variant':10s} " + " | ".join(f"d={d}: differing runs / rows off vs VALU / max|diff|" for d in (40, 80, 160)))
    for lib in libs:
        env = dict(os.environ, HUMANVID_HIP_LIB=lib, HV_DIAG_CHILD="1")
        out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")]
        name = os.path.basename(lib)[4:-3]
        if not line:
            print(f"{name:10s} FAILED: {out.stderr[-300:]}")
            continue
        r = json.loads(line[0][7:])
        print(f"{name:10s} " + " | ".join(
            f"{r[d]['runs_differing_from_run0']}/{REPS - 1}  {r[d]['rows_off_vs_valu']:7d}  {r[d]['max_abs_vs_valu']:.4f}"
            for d in ("40", "80", "160")))


if __name__ == "__main__":
    child() if os.environ.get("HV_DIAG_CHILD") else main()
