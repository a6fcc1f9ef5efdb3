"""ReferenceAttentionControl -- drop-in for /root/reference/src/models/mutual_self_attention.py:19-363.

The reference monkey-patches `.forward` of every transformer block; "write" mode banks the
post-norm1 features of the ReferenceNet, `update()` copies them (cast to fp16) onto the denoising
UNet's blocks, "read" mode concatenates them to the self-attention keys/values, with the
CFG-unconditional half recomputed without the bank.  Here the blocks are parameter containers and
the arithmetic lives in the native executors, so this class only does the book-keeping the
pipeline relies on: block discovery (same DFS + stable sort by hidden size, :267-287), the `bank`
lists on the modules, `update()` and `clear()`.  The executors pick the mode up from
`unet._reference_mode`.
"""
from __future__ import annotations

import itertools

import torch

from .unet3d import TemporalBasicTransformerBlock


_GENERATION = itertools.count(1)


def _bump(unet):
    """Every change of the banks of `unet` (new control, update, clear) gets a fresh, process-wide unique generation
    number; the native executor re-projects its bank K/V when the number it packed for differs.  (Keys derived from
    tensor addresses or `id()` are not safe: the caching allocator and CPython both hand freed addresses out again.)"""
    unet._reference_generation = next(_GENERATION)


def torch_dfs(model: torch.nn.Module):
    result = [model]
    for child in model.children():
        result += torch_dfs(child)
    return result


def _block_types():
    from .unet2d import BasicTransformerBlock

    return BasicTransformerBlock, TemporalBasicTransformerBlock


class ReferenceAttentionControl:
    def __init__(
        self,
        unet,
        mode="write",
        do_classifier_free_guidance=False,
        attention_auto_machine_weight=float("inf"),
        gn_auto_machine_weight=1.0,
        style_fidelity=1.0,
        reference_attn=True,
        reference_adain=False,
        fusion_blocks="midup",
        batch_size=1,
    ) -> None:
        self.unet = unet
        assert mode in ["read", "write"]
        assert fusion_blocks in ["midup", "full"]
        if reference_adain:
            raise NotImplementedError("reference_adain is not used by the CamAnimate pipelines")
        self.reference_attn = reference_attn
        self.reference_adain = reference_adain
        self.fusion_blocks = fusion_blocks
        self.mode = mode
        self.do_classifier_free_guidance = do_classifier_free_guidance
        if self.reference_attn:
            modules = self._modules(self.unet, both_kinds=True)
            for i, module in enumerate(modules):
                module.bank = []
                module.attn_weight = float(i) / float(len(modules))
            unet._reference_mode = dict(mode=mode, do_cfg=bool(do_classifier_free_guidance), fusion_blocks=fusion_blocks)
            _bump(unet)

    def _modules(self, unet, both_kinds=False, kind=None):
        basic, temporal = _block_types()
        kinds = (basic, temporal) if both_kinds else (kind,)
        if self.fusion_blocks == "midup":
            mods = torch_dfs(unet.mid_block) + torch_dfs(unet.up_blocks)
        else:
            mods = torch_dfs(unet)
        mods = [m for m in mods if isinstance(m, kinds)]
        return sorted(mods, key=lambda x: -x.norm1.normalized_shape[0])

    def update(self, writer, dtype=torch.float16):
        if self.reference_attn:
            basic, temporal = _block_types()
            readers = self._modules(self.unet, kind=temporal)
            writers = writer._modules(writer.unet, kind=basic)
            for r, w in zip(readers, writers):
                r.bank = [v.clone().to(dtype) for v in w.bank]
            _bump(self.unet)

    def clear(self):
        if self.reference_attn:
            for r in self._modules(self.unet, both_kinds=True):
                r.bank.clear()
            _bump(self.unet)
