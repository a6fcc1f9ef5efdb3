"""Loader for the product library libhumanvid_hip.so.  There is NO fallback: if the HIP library
was not built, or no MI355X/ROCm device is visible, the product path raises."""
from __future__ import annotations

import os

from ._abi import HvLibrary

_LIB = None
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libhumanvid_hip.so")


def load() -> HvLibrary:
    global _LIB
    if _LIB is None:
        override = os.environ.get("HUMANVID_HIP_LIB")  # A/B of build variants (tools/build_variant.sh); same C ABI
        if override:
            _LIB = HvLibrary(override)
            _apply_tuning(_LIB)
            return _LIB
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). humanvid_amd has no CPU or eager-PyTorch fallback."
            )
        _LIB = HvLibrary(LIB_PATH)
        _apply_tuning(_LIB)
    return _LIB


def _apply_tuning(lib):
    """HUMANVID_TUNING="key=value,key=value": hv_set_tuning calls applied at load time (same-box A/Bs of kernel selections
    under the tests and the bench; every selection computes the same function)."""
    for kv in filter(None, os.environ.get("HUMANVID_TUNING", "").split(",")):
        k, v = kv.split("=")
        lib.call("hv_set_tuning", int(k), int(v))


def require_gpu():
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError("humanvid_amd needs a ROCm GPU (MI355X); torch.cuda.is_available() is False")
    return torch.device("cuda", torch.cuda.current_device())


def current_stream() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream
