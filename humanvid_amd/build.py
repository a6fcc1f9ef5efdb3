"""Build libhumanvid_hip.so (gfx950) in-tree with hipcc; one translation unit per kernel family,
compiled in parallel.  Called by __graft_entry__.build(); the built .so travels to the GPU box."""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libhumanvid_hip.so")
UNITS = ["hv_api.cpp", "k_gemm.hip", "k_conv.hip", "k_norm.hip", "k_attention.hip", "k_temporal.hip",
         "k_elementwise.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(REPO, "include"), "-I" + CSRC]


def _newer(target, deps):
    return os.path.exists(target) and all(os.path.getmtime(target) >= os.path.getmtime(d) for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(os.path.join(LIBDIR, "obj"), exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(REPO, "include", "humanvid_hip.h"))
    jobs = []
    objs = []
    for u in UNITS:
        src = os.path.join(CSRC, u)
        obj = os.path.join(LIBDIR, "obj", u.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or not _newer(obj, [src] + headers):
            jobs.append([HIPCC, *FLAGS, "-x", "hip", "-c", src, "-o", obj])
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for cmd, res in zip(jobs, ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs)):
                if res.returncode != 0:
                    raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), res.stderr[-4000:]))
                if verbose and res.stderr:
                    print(res.stderr)
    if jobs or not os.path.exists(LIB):
        res = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB],
                             capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("link failed:\n" + res.stderr[-4000:])
    return LIB


def build_hwcheck() -> str:
    """tools/hwcheck.cpp: a C++ consumer of include/humanvid_hip.h with no Python in the process (seconds-long hardware
    check of hv_groupnorm_affine and the hv_gemm tile policies; run by tests/test_gpu_cabi.py) -> tools/bin/hwcheck"""
    src = os.path.join(REPO, "tools", "hwcheck.cpp")
    out = os.path.join(REPO, "tools", "bin", "hwcheck")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not _newer(out, [src, LIB, os.path.join(REPO, "include", "humanvid_hip.h")]):
        res = subprocess.run([HIPCC, "-O2", "-Wno-unused-value", "-Wno-unused-result", src, "-I" + os.path.join(REPO, "include"),
                              "-L" + LIBDIR, "-lhumanvid_hip", "-Wl,-rpath,$ORIGIN/../../humanvid_amd/lib", "-o", out],
                             capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hwcheck build failed:\n" + res.stderr[-4000:])
    return out


if __name__ == "__main__":
    print(build(force="--force" in os.sys.argv, verbose=True))
    print(build_hwcheck())
