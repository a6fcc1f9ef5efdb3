import glob, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch, kernel_cases as kc
from humanvid_amd import _abi, lib
libs = {"default": lib.LIB_PATH}
for p in sorted(glob.glob(os.path.join(REPO, "tools", "variants", "lib_*.so"))):
    libs[os.path.basename(p)[4:-3]] = p
for name, path in libs.items():
    L = _abi.HvLibrary(path)
    cx = kc.Ctx(L, "cuda", lib.current_stream())
    for (D, Lq, Lb, n) in [(40, 64, 64, 2), (40, 1536, 1536, 4), (40, 200, 72, 4)]:
        try:
            e = kc.case_attention(cx, D=D, n_img=n, Lq=Lq, Lb=Lb)
            print(name, D, Lq, Lb, "ok", e, flush=True)
        except AssertionError as ex:
            print(name, D, Lq, Lb, "FAIL", ex, flush=True)
