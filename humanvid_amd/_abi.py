"""ctypes mirror of include/humanvid_hip.h (the C ABI of libhumanvid_hip.so).

No torch, no numpy: plain pointers (ints) and sizes, exactly what the header declares.  The
mirror is verified at load time against `hv_struct_sizes()`.
"""
from __future__ import annotations

import ctypes as C

P = C.c_void_p
I = C.c_int
L = C.c_long
F = C.c_float

ACT_NONE, ACT_SILU, ACT_RELU = 0, 1, 2
CONV_S1, CONV_S2, CONV_UP2 = 0, 1, 2
HV_OK, HV_EINVAL, HV_ENOTSUP, HV_EHIP = 0, -1, -2, -3


class _S(C.Structure):
    def __init__(self, **kw):
        super().__init__()
        names = {n for n, _ in self._fields_}
        for k, v in kw.items():
            if k not in names:
                raise TypeError(f"{type(self).__name__}: unknown field {k}")
            setattr(self, k, v)


class GemmParams(_S):
    _fields_ = [
        ("X", P), ("ldx", L), ("X2", P), ("ldx2", L), ("K1", I),
        ("W", P), ("Y", P), ("ldy", L), ("out_f32", I),
        ("Yt", P), ("ldyt", L), ("n_split", I),
        ("M", I), ("N", I), ("K", I),
        ("pro_scale", P), ("pro_shift", P), ("rows_per_image", I), ("pro_act", I),
        ("bias", P), ("row_mean", P), ("row_rstd", P), ("colsum", P),
        ("pe", P), ("pe_period", I), ("pe_frames", I),
        ("rowvec", P), ("rowvec_period", I),
        ("residual", P), ("ldr", L),
        ("geglu", I), ("out_act", I),
        ("perm_x", I), ("perm_y", I), ("perm_p", I),
        ("gn_part", P), ("gn_rows_per_image", I), ("ln_part", P),
    ]


class Conv3x3Params(_S):
    _fields_ = [
        ("X", P), ("C1", I), ("X2", P), ("C2", I), ("W", P), ("Y", P),
        ("n_images", I), ("Hs", I), ("Ws", I), ("Ho", I), ("Wo", I), ("Cout", I),
        ("mode", I),
        ("pro_scale", P), ("pro_shift", P), ("pro_act", I),
        ("bias", P), ("rowvec", P), ("images_per_rowvec", I), ("rowvec_ld", L),
        ("residual", P), ("residual_images", I), ("out_act", I),
        ("gn_part", P),
    ]


class GroupNormParams(_S):
    _fields_ = [
        ("X", P), ("C1", I), ("X2", P), ("C2", I),
        ("n_images", I), ("pixels", I), ("groups", I), ("eps", F),
        ("gamma", P), ("beta", P), ("partial", P), ("splits", I),
        ("scale", P), ("shift", P),
    ]


class GnPartsParams(_S):
    _fields_ = [
        ("part1", P), ("parts1", I), ("C1", I), ("part2", P), ("parts2", I), ("C2", I),
        ("n_images", I), ("pixels", I), ("groups", I), ("eps", F),
        ("gamma", P), ("beta", P), ("scale", P), ("shift", P),
    ]


class AttentionParams(_S):
    _fields_ = [
        ("Q", P), ("ldq", L), ("K", P), ("ldk", L), ("Vt", P), ("ldvt", L),
        ("K2", P), ("ldk2", L), ("Vt2", P), ("ldvt2", L), ("bank_sel", P),
        ("O", P), ("ldo", L),
        ("n_images", I), ("heads", I), ("D", I), ("Lq", I), ("L1", I), ("L2", I),
        ("scale", F),
    ]


class TemporalAttentionParams(_S):
    _fields_ = [
        ("Q", P), ("ldq", L), ("K", P), ("V", P), ("ldkv", L),
        ("kv_stride_b", L), ("kv_stride_chunk", L), ("kv_chunk", I),
        ("O", P), ("ldo", L),
        ("B", I), ("Fq", I), ("Fkv", I), ("P", I), ("heads", I), ("D", I), ("scale", F), ("qo_chunked", I),
    ]


_STRUCTS = (GemmParams, Conv3x3Params, GroupNormParams, AttentionParams, TemporalAttentionParams, GnPartsParams)

# every symbol include/humanvid_hip.h declares: name -> (restype, argtypes)
PROTOTYPES = {
    "hv_last_error": (C.c_char_p, []),
    "hv_abi_version": (I, []),
    "hv_struct_sizes": (I, [C.POINTER(I), I]),
    "hv_gemm": (I, [C.POINTER(GemmParams), P]),
    "hv_conv3x3": (I, [C.POINTER(Conv3x3Params), P]),
    "hv_groupnorm_affine": (I, [C.POINTER(GroupNormParams), P]),
    "hv_gemm_gn_parts": (I, [C.POINTER(GemmParams)]),
    "hv_gemm_ln_parts": (I, [C.POINTER(GemmParams)]),
    "hv_layernorm_from_parts": (I, [P, I, I, I, F, P, P, P]),
    "hv_conv3x3_gn_parts": (I, [C.POINTER(Conv3x3Params)]),
    "hv_groupnorm_from_parts": (I, [C.POINTER(GnPartsParams), P]),
    "hv_layernorm_stats": (I, [P, L, I, I, F, P, P, P]),
    "hv_attention": (I, [C.POINTER(AttentionParams), P]),
    "hv_attention_fp8_quantize": (I, [P, L, P, L, I, I, I, I, P, P, P, P, L, P, L, I, P]),
    "hv_attention_fp8": (I, [C.POINTER(AttentionParams), P, P, P, P, P]),
    "hv_set_tuning": (I, [I, I]),
    "hv_temporal_attention": (I, [C.POINTER(TemporalAttentionParams), P]),
    "hv_pack_ncfhw": (I, [P, I, I, I, I, I, I, P, I, I, P, I, P]),
    "hv_unpack_nhwc": (I, [P, I, I, I, I, I, I, P, I, P]),
    "hv_pixel_unshuffle": (I, [P, I, I, I, I, I, I, P, P]),
    "hv_plucker_unshuffle": (I, [P, P, I, I, I, I, P, P]),
    "hv_affine_apply": (I, [P, L, I, I, I, P, P, I, P, L, P]),
    "hv_affine_apply_cat": (I, [P, L, I, P, L, I, I, I, P, P, I, P, L, P]),
    "hv_timestep_embedding": (I, [P, I, I, P, P]),
    "hv_accumulate_window": (I, [P, I, I, I, I, I, I, P, I, P, P, P]),
    "hv_cfg_ddim_step": (I, [P, P, P, I, I, I, I, I, P, P]),
    "hv_graph_begin": (I, [P]),
    "hv_graph_end": (I, [P, C.POINTER(P)]),
    "hv_graph_launch": (I, [P, P]),
    "hv_graph_destroy": (I, [P]),
    "hv_cmdlist_begin": (I, []),
    "hv_cmdlist_cut": (I, [C.POINTER(P)]),
    "hv_cmdlist_end": (I, [C.POINTER(P)]),
    "hv_cmdlist_size": (I, [P]),
    "hv_cmdlist_run": (I, [P, P]),
    "hv_cmdlist_fallbacks": (I, []),
    "hv_cmdlist_destroy": (I, [P]),
    "hv_profile_begin": (I, []),
    "hv_profile_end": (I, [C.c_char_p, I]),
    "hv_event_create": (I, [C.POINTER(P)]),
    "hv_event_record": (I, [P, P]),
    "hv_event_elapsed_ms": (I, [P, P, C.POINTER(F)]),
    "hv_event_destroy": (I, [P]),
}


class HvError(RuntimeError):
    pass


class HvLibrary:
    """A loaded C-ABI library with typed entry points; raises like the reference's Python would
    (ValueError for bad shapes, NotImplementedError for unsupported geometry, RuntimeError for HIP)."""

    def __init__(self, path: str):
        self.path = path
        self.cdll = C.CDLL(path)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(self.cdll, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        sizes = (I * 16)()
        n = self.cdll.hv_struct_sizes(sizes, 16)
        mine = [C.sizeof(s) for s in _STRUCTS]
        if n != len(mine) or list(sizes[:n]) != mine:
            raise HvError(f"ABI mismatch between {path} and humanvid_amd/_abi.py: {list(sizes[:n])} vs {mine}")

    def check(self, rc: int):
        if rc == HV_OK:
            return
        msg = self.cdll.hv_last_error().decode()
        if rc == HV_EINVAL:
            raise ValueError(msg)
        if rc == HV_ENOTSUP:
            raise NotImplementedError(msg)
        raise HvError(msg)

    def call(self, name: str, *args):
        self.check(getattr(self.cdll, name)(*args))
