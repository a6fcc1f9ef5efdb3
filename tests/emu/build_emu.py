"""Build the host-emulated copy of the kernels (tests only; see tests/emu/hv_emu.h)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_build", "libhumanvid_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def build(force=False):
    srcdir = os.path.join(REPO, "humanvid_amd", "csrc")
    deps = [os.path.join(srcdir, f) for f in os.listdir(srcdir)] + [
        os.path.join(HERE, "hv_emu.h"), os.path.join(REPO, "include", "humanvid_hip.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cxx = CLANG if os.path.exists(CLANG) else "clang++"
    cmd = [cxx, "-DHV_EMU", "-DHV_SINGLE_TU", "-x", "c++", "-O2", "-Wno-psabi", "-std=c++17", "-I" + os.path.join(REPO, "include"), "-I" + srcdir, "-I" + HERE,
           "-include", os.path.join(HERE, "hv_emu.h"), "-shared", "-fPIC", os.path.join(srcdir, "hv_api.cpp"),
           "-o", OUT]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
