#!/usr/bin/env python3
"""Scan hipcc's gfx950 assembly of the kernel translation units for instructions that write a VGPR which an MFMA issued
one or two instructions earlier reads as SrcA / SrcB (see the note on HV_MFMA_GUARD in humanvid_amd/csrc/hv_temporal.h).

usage: python tools/isa_scan.py            (compiles the units with -save-temps into a temporary directory)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNITS = ["k_gemm", "k_conv", "k_attention", "k_temporal"]


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def main():
    tmp = tempfile.mkdtemp(prefix="hv_isa_")
    for u in UNITS:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(REPO, "include"),
                        "-I" + os.path.join(REPO, "humanvid_amd", "csrc"), "-c", "-x", "hip",
                        os.path.join(REPO, "humanvid_amd", "csrc", u + ".hip"), "-o", os.path.join(tmp, u + ".o"), "-save-temps=obj"],
                       cwd=tmp, capture_output=True)
    for f in sorted(glob.glob(os.path.join(tmp, "*gfx950.s"))):
        kern, code = "?", []
        for raw in open(f):
            line = raw.strip()
            m = re.match(r"^(_Z\w+):", line)
            if m:
                kern = m.group(1)
                continue
            if not line or line[0] in ";." or re.match(r"^[\w.$]+:", line):
                continue
            code.append((kern, line))
        hits = {}
        for i, (k, line) in enumerate(code):
            if not line.startswith("v_mfma"):
                continue
            ops = [t.strip() for t in line.split(None, 1)[1].split(",")]
            src = regs(ops[1]) | regs(ops[2])
            for j in (1, 2):
                if i + j >= len(code):
                    break
                nxt = code[i + j][1]
                if nxt.startswith("v_mfma") or nxt.startswith("v_cmp") or not nxt.startswith("v_"):
                    continue
                if regs(nxt.split(None, 1)[1].split(",")[0].strip()) & src:
                    hits.setdefault(k, []).append((j, line[:70], nxt[:60]))
        print(os.path.basename(f).split("-hip")[0])
        for k, v in hits.items():
            print(f"  {k[:90]}: {len(v)} (e.g. +{v[0][0]}: {v[0][1]} | {v[0][2]})")
        if not hits:
            print("  none")


if __name__ == "__main__":
    sys.exit(main())
