"""Native executor for UNet3DConditionModel.forward on MI355X.

Restates the *schedule* of /root/reference/src/models/unet_3d.py:397-577 (and the block forwards it
calls: unet_3d_blocks.py:398-464, 540-583, 269-293, 682-746, 816-863; resnet.py:215-245;
transformer_3d.py:103-169; mutual_self_attention.py:147-228 read mode; motion_module.py:146-259,
351-388) as a sequence of launches into libhumanvid_hip.so.  Canonical layout: bf16,
[(b f)][h][w][c] == [(b f)][token][c]; every rearrange / cat / upsample / norm-apply of the
reference is absorbed into kernel addressing, prologues or epilogues:

  ResnetBlock3D  = gn-merge, GN apply + SiLU pass, conv3x3(+bias +temb), gn-merge, [1x1 shortcut
                   GEMM], GN apply + SiLU pass, conv3x3(+bias +residual)
  Transformer3D  = gn-merge, GN apply pass, GEMM proj_in, ln-merge, GEMM qkv(LN folded; V stored
                   transposed), flash attention (+bank keys for the conditional half),
                   GEMM out(+bias +folded cross-attention constant +residual), ln-merge,
                   GEMM ff1(LN folded, GEGLU), GEMM ff2(+residual), GEMM proj_out(+residual)
  motion module  = gn-merge, GN apply pass, GEMM proj_in, 2x[ln-merge, GEMM qkv(LN + positional
                   encoding folded), temporal attention, GEMM out(+residual)], ln-merge, GEMM ff1,
                   GEMM ff2(+residual), GEMM proj_out(+residual)
(gn-/ln-merge: the normalisation statistics come from partial sums that the PRODUCING kernel left in
its epilogue -- hv_groupnorm_from_parts / hv_layernorm_from_parts read kilobytes, not activations.)

The 1-key CLIP cross-attention is algebraically a per-batch constant (softmax over one key == 1):
to_out(to_v(e)) + bias, computed once per clip and added in the attn1 out-projection epilogue.
With a FrameShard (runner.py) the images are sharded along the frame axis; around every temporal
attention q | k | v are re-sharded frames <-> pixels by an all-to-all over RCCL (default), or the K/V
of all frames are all-gathered (HUMANVID_TEMPORAL_EXCHANGE=allgather) -- DESIGN.md section 5.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch

from . import _abi as A
from . import lib as hvlib
from . import ops, packing
from .runner import FrameShard, Runner, Workspace
from .unet3d import TemporalBasicTransformerBlock, UNet3DConditionModel, transformer_locations

BF16 = torch.bfloat16
F32 = torch.float32


class UNet3DEngine:
    def __init__(self, unet, device=None, shard: Optional[FrameShard] = None, kind: str = "denoise"):
        """kind="denoise": UNet3DConditionModel (read mode); kind="reference": the 2-D ReferenceNet, run
        with one frame per batch entry, which materialises the post-norm1 features as banks."""
        self.unet = unet
        self.kind = kind
        self.cfg = unet.config if kind == "denoise" else unet._walk_cfg
        self.written_banks: Dict[str, torch.Tensor] = {}
        self.device = device or hvlib.require_gpu()
        self.lib = hvlib.load()
        self.ws = Workspace(self.device)
        self.shard = shard
        self.heads = self.cfg["attention_head_dim"]
        self.groups = self.cfg["norm_num_groups"]
        self.eps = self.cfg["norm_eps"]
        self.w: Dict[str, torch.Tensor] = {}
        self.bank_kv: Dict[str, tuple] = {}
        self.bank_version = None
        self.cross_const: Dict[str, torch.Tensor] = {}
        self.do_cfg = True
        # BASELINE.json configs[4]: spatial attention on the fp8 (e4m3) MFMA -- hv_attention_fp8 (transposed-V kernel family)
        # (denoising UNet only: the ReferenceNet write pass runs once per clip and keeps its banks at bf16 precision)
        self.attn_fp8 = os.environ.get("HUMANVID_ATTENTION_FP8", "0") == "1" and kind == "denoise"
        self.bank_fp8 = {}
        self._sel_cache: Dict[tuple, torch.Tensor] = {}
        # optional observer `tap(name, activation [(b f),h,w,c])` called after every resnet / spatial transformer / motion
        # module, named by the reference's module path (e.g. 'down_blocks.0.attentions.1').  Buffers are reused and updated in place: the observer
        # must copy what it wants to keep.  Used by the parity tests; None in production (and under graph capture).
        self.tap = None
        self.tap_fine = None  # a second observer for the intermediate activations (up path, mid block, resampling convolutions)
        # CFG half this executor instance evaluates when it is given one batch entry (B = 1) of a guided step: None = whole
        # batch ([unconditional | conditional] in one forward), 0 / 1 = that half only (clone_for_half: the two halves of a
        # frame-sharded step run on two streams so that one half's temporal exchange hides under the other's kernels)
        self.cfg_half = None
        self._pack()
        mmk = self.cfg.get("motion_module_kwargs") or {}
        self.run = Runner(self.device, self.w, self.ws, self.groups, shard,
                          temporal_heads=mmk.get("num_attention_heads", 8))

    # ------------------------------------------------------------------------------------------
    @property
    def stream(self) -> int:
        return hvlib.current_stream()

    def clone_for_half(self, half: int, ws: "Workspace | None" = None) -> "UNet3DEngine":
        """An executor for ONE CFG half (batch entry `half` of [unconditional, conditional]) with its own workspace and
        statistics tables, sharing the packed weights, the projected reference banks and the folded cross-attention
        constants of this one (read-only during a step).  Make the clones after the banks / embeddings are set.
        `ws`: a workspace kept by the caller across calls (the pipeline caches one per half and stream, so that repeated
        denoise() calls do not re-allocate gigabytes on freshly drawn side streams)."""
        import copy

        e = copy.copy(self)
        e.ws = ws if ws is not None else Workspace(self.device)
        mmk = self.cfg.get("motion_module_kwargs") or {}
        e.run = Runner(self.device, self.w, e.ws, self.groups, self.shard, temporal_heads=mmk.get("num_attention_heads", 8))
        e._sel_cache = {}
        e.cfg_half = int(half)
        e.tap = None
        return e

    def _dev(self, t: torch.Tensor, dtype=None) -> torch.Tensor:
        return t.detach().to(device=self.device, dtype=dtype or t.dtype).contiguous()

    def _pack(self):
        """Derive the device-side operands from the parameter containers (once per weight load)."""
        sd = {k: v for k, v in self.unet.state_dict().items()}
        w = self.w
        d = self._dev

        def gn(p):
            w[p + ".g"], w[p + ".b"] = d(sd[p + ".weight"], F32), d(sd[p + ".bias"], F32)

        def conv(p, cin_pad=None):
            w[p + ".w"] = d(packing.pack_conv3x3(sd[p + ".weight"].float(), cin_pad))
            w[p + ".bias"] = d(sd[p + ".bias"], F32)

        def lin(p, bias=True):
            w[p + ".w"] = d(packing.pack_linear(sd[p + ".weight"].float()))
            if bias and (p + ".bias") in sd:
                w[p + ".bias"] = d(sd[p + ".bias"], F32)

        def lnfold(name, wcat, bias, norm, geglu=False, pe=None):
            gamma, beta = sd[norm + ".weight"].float(), sd[norm + ".bias"].float()
            wcat = wcat.to(gamma.device)
            wf, colsum, bf = packing.fold_layernorm(wcat.float(), None if bias is None else bias.float(), gamma, beta)
            pet = packing.pe_table(pe, wcat) if pe is not None else None
            if geglu:
                order = packing.geglu_row_order(wf.shape[0])
                wf, colsum, bf = wf[order], colsum[order], bf[order]
            w[name + ".w"], w[name + ".colsum"], w[name + ".bias"] = d(wf), d(colsum, F32), d(bf, F32)
            if pet is not None:
                w[name + ".pe"] = d(pet, F32)

        cin = self.cfg["in_channels"]
        conv("conv_in", packing.round_up(cin, 32))
        lin("time_embedding.linear_1")
        lin("time_embedding.linear_2")
        temb_w, temb_b, self.temb_off = [], [], {}
        off = 0
        for spec in self.unet.specs:
            p = spec.prefix
            for j, (m, s, o) in enumerate(spec.resnets):
                r = f"{p}.resnets.{j}"
                gn(r + ".norm1")
                conv(r + ".conv1")
                gn(r + ".norm2")
                conv(r + ".conv2")
                if (r + ".conv_shortcut.weight") in sd:
                    lin(r + ".conv_shortcut")
                temb_w.append(sd[r + ".time_emb_proj.weight"].float())
                temb_b.append(sd[r + ".time_emb_proj.bias"].float())
                self.temb_off[r] = off
                off += o
            n_attn = len(spec.resnets) - 1 if spec.kind == "mid" else len(spec.resnets)
            for j in range(n_attn):
                if spec.has_attn:
                    a = f"{p}.attentions.{j}"
                    t = a + ".transformer_blocks.0"
                    gn(a + ".norm")
                    lin(a + ".proj_in")
                    lin(a + ".proj_out")
                    qkv = torch.cat([sd[t + f".attn1.to_{x}.weight"] for x in "qkv"], dim=0)
                    lnfold(t + ".qkv", qkv, None, t + ".norm1")
                    if self.kind == "reference":  # LayerNorm itself as a GEMM with W = I (bank = norm1(h))
                        lnfold(t + ".norm1_id", torch.eye(qkv.shape[1]), None, t + ".norm1")
                    w[t + ".bank_kv.w"] = d(packing.pack_linear(torch.cat(
                        [sd[t + ".attn1.to_k.weight"], sd[t + ".attn1.to_v.weight"]], dim=0).float()))
                    lin(t + ".attn1.to_out.0")
                    lin(t + ".attn2.to_v", bias=False)
                    lin(t + ".attn2.to_out.0")
                    lnfold(t + ".ff1", sd[t + ".ff.net.0.proj.weight"], sd[t + ".ff.net.0.proj.bias"], t + ".norm3",
                           geglu=True)
                    lin(t + ".ff.net.2")
                if spec.has_motion:
                    mm = f"{p}.motion_modules.{j}.temporal_transformer"
                    gn(mm + ".norm")
                    lin(mm + ".proj_in")
                    lin(mm + ".proj_out")
                    nb = self.cfg["motion_module_kwargs"].get("num_transformer_block", 1)
                    for li in range(nb):
                        b = f"{mm}.transformer_blocks.{li}"
                        na = len(self.cfg["motion_module_kwargs"]["attention_block_types"])
                        for ai in range(na):
                            ab = f"{b}.attention_blocks.{ai}"
                            qkv = torch.cat([sd[ab + f".to_{x}.weight"] for x in "qkv"], dim=0)
                            pe = sd.get(ab + ".pos_encoder.pe")
                            lnfold(ab + ".qkv", qkv, None, f"{b}.norms.{ai}", pe=None if pe is None else pe[0].float())
                            lin(ab + ".to_out.0")
                        lnfold(b + ".ff1", sd[b + ".ff.net.0.proj.weight"], sd[b + ".ff.net.0.proj.bias"],
                               b + ".ff_norm", geglu=True)
                        lin(b + ".ff.net.2")
            if spec.kind == "down" and spec.resample:
                conv(f"{p}.downsamplers.0.conv")
            if spec.kind == "up" and spec.resample:
                conv(f"{p}.upsamplers.0.conv")
        w["temb_all.w"] = d(packing.pack_linear(torch.cat(temb_w, dim=0)))
        w["temb_all.bias"] = d(torch.cat(temb_b, dim=0), F32)
        self.temb_total = off
        if self.kind == "denoise":
            gn("conv_norm_out")
            conv("conv_out")
            self.locations = transformer_locations(self.unet)
        else:
            self.locations = [f"{s.prefix}.attentions.{j}" for s in self.unet.specs if s.has_attn
                              for j in range(len(s.resnets) - 1 if s.kind == "mid" else len(s.resnets))]

    # ------------------------------------------------------------------------------------------
    def set_encoder_hidden_states(self, ehs: torch.Tensor):
        """ehs [B, 1, 768]: fold the one-key cross-attention of every transformer into a per-batch
        constant (attention.py:410-424 -> to_out(to_v(e)) + bias)."""
        assert ehs.ndim == 3 and ehs.shape[1] == 1, "the CamAnimate path feeds exactly one CLIP token"
        B = ehs.shape[0]
        e = self._dev(ehs[:, 0], BF16)
        st = self.stream
        for loc in self.locations:
            t = loc + ".transformer_blocks.0"
            C = self.w[t + ".attn2.to_out.0.w"].shape[0]
            v = self.ws.get("xc_v", (B, C))
            ops.gemm(self.lib, st, e, self.w[t + ".attn2.to_v.w"], v)
            c = self.cross_const.get(loc)
            if c is None or c.shape[0] != B:
                c = torch.empty(B, C, dtype=F32, device=self.device)
                self.cross_const[loc] = c
            ops.gemm(self.lib, st, v, self.w[t + ".attn2.to_out.0.w"], c, bias=self.w[t + ".attn2.to_out.0.bias"])

    def set_reference_banks(self, banks: Optional[Dict[str, torch.Tensor]], do_cfg: bool):
        """banks: {transformer location -> [b, Nb, C]} (norm1 features of the ReferenceNet write pass,
        already rounded through fp16 by update(), mutual_self_attention.py:338).  Projects bank keys /
        transposed values once per clip."""
        self.do_cfg = do_cfg
        self.bank_kv = {}
        self.bank_fp8 = {}
        if not banks:
            return
        st = self.stream
        for loc in self.locations:
            bank = banks.get(loc)
            if bank is None:
                continue
            b, Nb, C = bank.shape
            x = self._dev(bank.reshape(b * Nb, C), BF16)
            k2 = torch.empty(b * Nb, C, dtype=BF16, device=self.device)
            vt2 = torch.empty(C, b * Nb, dtype=BF16, device=self.device)
            ops.gemm(self.lib, st, x, self.w[loc + ".transformer_blocks.0.bank_kv.w"], k2, yt=vt2, n_split=C, ldy=C)
            self.bank_kv[loc] = (k2, vt2, b, Nb)
            if self.attn_fp8:  # e4m3: K scales / V amax of the bank once per clip (the quantised copies are rewritten per
                # attention call, with the V scale the call's own values share -- hv_attention_fp8_quantize)
                ks2 = torch.empty(b, self.heads, (Nb + 63) // 64, dtype=F32, device=self.device)
                va2 = torch.empty(self.heads, dtype=F32, device=self.device)
                ops.attention_fp8_quantize(self.lib, st, k2, vt2, ks2, va2, n_images=b, heads=self.heads, D=C // self.heads,
                                           L=Nb, ldk=C, ldvt=b * Nb, phase=1)
                k28 = torch.empty(b * Nb, C, dtype=torch.uint8, device=self.device)
                vt28 = torch.empty(C, b * Nb, dtype=torch.uint8, device=self.device)
                self.bank_fp8[loc] = (ks2, va2, k28, vt28)

    def _banks_from_modules(self):
        """Pick up banks installed on the transformer blocks by ReferenceAttentionControl.update()."""
        mode = self.unet._reference_mode
        blocks = {n.rsplit(".transformer_blocks.0", 1)[0]: m for n, m in self.unet.named_modules()
                  if isinstance(m, TemporalBasicTransformerBlock)}
        # generation number bumped by ReferenceAttentionControl on every __init__ / update() / clear(); plus which blocks
        # currently hold a bank (covers banks assigned by hand between two control calls)
        held = tuple(loc for loc, m in sorted(blocks.items()) if m.bank)
        key = (getattr(self.unet, "_reference_generation", 0), held, None if mode is None else mode.get("do_cfg"))
        if key == self.bank_version:
            return
        self.bank_version = key
        if mode is None or mode.get("mode") != "read":
            self.set_reference_banks(None, do_cfg=False)
            return
        banks = {loc: m.bank[0] for loc, m in blocks.items() if m.bank}
        self.set_reference_banks(banks, do_cfg=bool(mode.get("do_cfg")))

    # ------------------------------------------------------------------------------------------
    def forward_ncfhw(self, sample, timestep, encoder_hidden_states, pose_cond_fea):
        """The reference's tensor interface: sample [b,c,f,h,w] -> [b,c_out,f,h,w] (same dtype)."""
        if sample.device.type != "cuda":
            raise RuntimeError("UNet3DConditionModel.forward needs CUDA/HIP tensors: there is no CPU path")
        b, c, f, h, w = sample.shape
        up = 2 ** self.unet.num_upsamplers
        if h % up or w % up:
            raise NotImplementedError(f"latent size must be a multiple of {up} (forward_upsample_size is not supported)")
        self._banks_from_modules()
        # the folded cross-attention constants are recomputed on every call of the tensor interface (32 tiny GEMMs): a
        # cache keyed on the embedding's address would go stale when the allocator reuses the address for the next clip
        self.set_encoder_hidden_states(encoder_hidden_states)
        st = self.stream
        src = sample if sample.dtype in (F32, BF16) else sample.float()
        x_in = self.ws.get("x_in", (b * f, h, w, packing.round_up(c, 32)))
        ops.pack_ncfhw(self.lib, st, src.contiguous(), x_in, rep=1)
        cond = None
        if pose_cond_fea is not None:
            pc = pose_cond_fea if pose_cond_fea.dtype in (F32, BF16) else pose_cond_fea.float()
            cond = self.ws.get("cond_in", (b * f, h, w, pose_cond_fea.shape[1]))
            ops.pack_ncfhw(self.lib, st, pc.contiguous(), cond, rep=1)
        t = torch.as_tensor(timestep, device=self.device).to(F32).reshape(-1)
        t = t.expand(b).contiguous() if t.numel() == 1 else t
        y = self.forward_nhwc(x_in, t, cond, B=b, F=f)
        out = torch.empty(b, self.cfg["out_channels"], f, h, w, dtype=src.dtype, device=self.device)
        ops.unpack_nhwc(self.lib, st, y, out)
        return out.to(sample.dtype)

    def forward_reference(self, sample, timestep, encoder_hidden_states):
        """ReferenceNet pass: sample [b,4,h,w] -> last hidden state [b,C0,h,w]; fills `bank` of every
        BasicTransformerBlock when the model is in write mode (pipeline_pose2vid_long.py:470-480)."""
        if sample.device.type != "cuda":
            raise RuntimeError("UNet2DConditionModel.forward needs CUDA/HIP tensors: there is no CPU path")
        b, c, h, w_ = sample.shape
        self.set_encoder_hidden_states(encoder_hidden_states)  # every call: see forward_ncfhw
        self.bank_kv = {}
        st = self.stream
        src = sample if sample.dtype in (F32, BF16) else sample.float()
        x_in = self.ws.get("x_in", (b, h, w_, packing.round_up(c, 32)))
        ops.pack_ncfhw(self.lib, st, src.reshape(b, c, 1, h, w_).contiguous(), x_in, rep=1)
        t = torch.as_tensor(timestep, device=self.device).to(F32).reshape(-1)
        t = t.expand(b).contiguous() if t.numel() == 1 else t
        self.written_banks = {}
        x = self.forward_nhwc(x_in, t, None, B=b, F=1)
        mode = getattr(self.unet, "_reference_mode", None)
        if mode is not None and mode.get("mode") == "write":
            from .unet2d import BasicTransformerBlock

            blocks = {n.rsplit(".transformer_blocks.0", 1)[0]: m for n, m in self.unet.named_modules()
                      if isinstance(m, BasicTransformerBlock)}
            for loc, bank in self.written_banks.items():
                if hasattr(blocks[loc], "attn_weight"):  # registered by the control (fusion_blocks filter)
                    blocks[loc].bank.append(bank.clone())
        out = torch.empty(b, x.shape[3], 1, h, w_, dtype=src.dtype, device=self.device)
        ops.unpack_nhwc(self.lib, st, x, out)
        return out[:, :, 0].to(sample.dtype)

    # ------------------------------------------------------------------------------------------
    def forward_nhwc(self, x_in: torch.Tensor, t_dev: torch.Tensor, cond: Optional[torch.Tensor], B: int, F: int):
        """x_in [(B F), h, w, 32] bf16 (4 real channels); t_dev fp32 [B] on device; cond
        [(B F) or F, h, w, C0] bf16 or None.  F is the LOCAL frame count when sharded.
        Returns conv_out output [(B F), h, w, out_channels] bf16."""
        L, st, w, ws = self.lib, self.stream, self.w, self.ws
        n, H, W, _ = x_in.shape
        assert n == B * F
        boc = tuple(self.cfg["block_out_channels"])
        G = self.groups

        # ---- time embedding: sinusoid -> MLP -> all 22 resnet projections in one GEMM
        te = ws.get("te_in", (B, boc[0]))
        ops.timestep_embedding(L, st, t_dev, te)
        e1 = ws.get("te_1", (B, boc[0] * 4))
        ops.gemm(L, st, te, w["time_embedding.linear_1.w"], e1, bias=w["time_embedding.linear_1.bias"])
        emb = ws.get("te_2", (B, boc[0] * 4))
        ops.gemm(L, st, e1, w["time_embedding.linear_2.w"], emb, bias=w["time_embedding.linear_2.bias"],
                 pro_act=A.ACT_SILU)
        temb = ws.get("temb_all", (B, self.temb_total), F32)
        ops.gemm(L, st, emb, w["temb_all.w"], temb, bias=w["temb_all.bias"], pro_act=A.ACT_SILU)

        gn_affine, ln_stats, feed_forward = self.run.gn_affine, self.run.ln_stats, self.run.feed_forward
        # every activation a GroupNorm reads is written by a 3x3 convolution or by a projection-out GEMM: both leave the
        # GroupNorm partial statistics of what they store (Runner.conv_with_stats / gemm_with_stats), so gn_affine needs no
        # pass over the activation.  The statistics belong to THIS forward's buffers: start from an empty table.
        self.run.gn_parts.clear()
        self.run.ln_parts.clear()
        conv = self.run.conv_with_stats
        sharded = self.shard is not None and self.shard.active

        def resnet(prefix, x, skip, out_name):
            """ResnetBlock3D.forward, resnet.py:215-245 (skip = second concat source or None)."""
            _, h, ww, _ = x.shape
            cout = w[prefix + ".conv1.bias"].shape[0]
            sc, sh = gn_affine(x, prefix + ".norm1", self.eps, skip)
            h1 = ws.get(f"res_h1_{h}x{ww}x{cout}", (n, h, ww, cout))
            off = self.temb_off[prefix]
            conv(x, w[prefix + ".conv1.w"], h1, x2=skip, pro_scale=sc, pro_shift=sh, pro_act=A.ACT_SILU,
                 bias=w[prefix + ".conv1.bias"], rowvec=temb[:, off:], images_per_rowvec=F, rowvec_ld=self.temb_total)
            if (prefix + ".conv_shortcut.w") in w:
                res = ws.get(f"res_sc_{h}x{ww}x{cout}", (n, h, ww, cout))
                c1 = x.shape[3]
                ops.gemm(L, st, x.view(-1, c1), w[prefix + ".conv_shortcut.w"], res.view(-1, cout),
                         x2=None if skip is None else skip.view(-1, skip.shape[3]), k1=c1,
                         bias=w[prefix + ".conv_shortcut.bias"])
            else:
                assert skip is None
                res = x
            sc, sh = gn_affine(h1, prefix + ".norm2", self.eps)
            out = ws.get(out_name, (n, h, ww, cout))
            conv(h1, w[prefix + ".conv2.w"], out, pro_scale=sc, pro_shift=sh, pro_act=A.ACT_SILU,
                 bias=w[prefix + ".conv2.bias"], residual=res)
            return out

        def proj_in(x2d, sc, sh, N, wt, bias, hid):
            """GroupNorm apply + proj_in: the normalisation as its own HBM-bound pass (hv_affine_apply) into a scratch
            activation, then the projection on the LDS-DMA GEMM kernel (round 1 fused the apply into the A-operand staging of
            the register-staged GEMM: 8.5 % MFMA-busy).  Folding the scale into per-image weights instead was rejected in
            round 3 on numerical grounds: sum_k W a x and sum_k W b cancel when a channel's |mean| >> std, which amplifies the
            bf16 rounding of W a by that ratio."""
            Mr, Cc = x2d.shape
            xn = ws.get(f"tr_n_{Mr}x{Cc}", (Mr, Cc))
            ops.affine_apply(L, st, x2d, sc, sh, xn, rows_per_image=N)
            self.run.gemm_ln(xn, wt, hid, bias=bias)

        def transformer(prefix, x):
            """Transformer3DModel.forward (transformer_3d.py:103-169) + patched block (read mode)."""
            _, h, ww, C = x.shape
            N = h * ww
            M = n * N
            t = prefix + ".transformer_blocks.0"
            x2d = x.view(M, C)
            sc, sh = gn_affine(x, prefix + ".norm", 1e-6)
            hid = ws.get(f"tr_h_{M}x{C}", (M, C))
            proj_in(x2d, sc, sh, N, w[prefix + ".proj_in.w"], w[prefix + ".proj_in.bias"], hid)
            mean, rstd = ln_stats(hid)
            if self.kind == "reference":  # "write" mode: bank.append(norm_hidden_states.clone())
                bank = ws.get("bank." + prefix, (M, C))
                ops.gemm(L, st, hid, w[t + ".norm1_id.w"], bank, bias=w[t + ".norm1_id.bias"], row_mean=mean,
                         row_rstd=rstd, colsum=w[t + ".norm1_id.colsum"])
                self.written_banks[prefix] = bank.view(n, N, C)
            qk = ws.get(f"tr_qk_{M}x{C}", (M, 2 * C))
            vt = ws.get(f"tr_vt_{M}x{C}", (C, M))  # values arrive transposed: written by the QKV GEMM epilogue
            ops.gemm(L, st, hid, w[t + ".qkv.w"], qk, bias=w[t + ".qkv.bias"], row_mean=mean, row_rstd=rstd,
                     colsum=w[t + ".qkv.colsum"], yt=vt, n_split=2 * C)
            ldqk, ldvt = 2 * C, M
            o = ws.get(f"tr_o_{M}x{C}", (M, C))
            kw = {}
            bank = self.bank_kv.get(prefix)
            if bank is not None:
                k2, vt2, bb, Nb = bank
                if self.do_cfg:
                    cond = 1 if bb > 1 else 0  # a one-entry bank (ReferenceNet run on the conditional embedding only)
                    if B == 2:
                        sel = [-1] * F + [cond] * (n - F)
                    else:  # one batch entry: the unconditional half, unless this executor evaluates the conditional one
                        sel = [cond] * n if self.cfg_half == 1 else [-1] * n
                else:
                    sel = [i // F if bb > 1 else 0 for i in range(n)]
                skey = (n, F, int(self.do_cfg), bb, self.cfg_half)
                sel_t = self._sel_cache.get(skey)
                if sel_t is None:  # uploaded once (keeps the step free of host copies / graph-capturable)
                    sel_t = torch.tensor(sel, dtype=torch.int32).to(self.device)
                    self._sel_cache[skey] = sel_t
                kw = dict(k2=k2, vt2=vt2, ldk2=C, ldvt2=bb * Nb, L2=Nb, bank_sel=sel_t)
            if self.attn_fp8:
                Dh = C // self.heads
                ks1 = ws.get(f"tr_ks_{n}x{N}", (n, self.heads, (N + 63) // 64), F32)
                va1 = ws.get("tr_va", (self.heads,), F32)
                k8 = ws.get(f"tr_k8_{M}x{C}", (M, C), torch.uint8)
                vt8 = ws.get(f"tr_vt8_{M}x{C}", (C, M), torch.uint8)
                own = dict(n_images=n, heads=self.heads, D=Dh, L=N, ldk=ldqk, ldvt=ldvt)
                ops.attention_fp8_quantize(L, st, qk[:, C:], vt, ks1, va1, phase=1, **own)
                kw8 = dict(kw)
                if bank is not None:
                    k2, vt2, bb, Nb = bank
                    ks2, va2, k28, vt28 = self.bank_fp8[prefix]
                    ops.attention_fp8_quantize(L, st, k2, vt2, ks2, va2, n_images=bb, heads=self.heads, D=Dh, L=Nb, ldk=C,
                                               ldvt=bb * Nb, phase=2, vfloor=va1, k8=k28, vt8=vt28)
                    ops.attention_fp8_quantize(L, st, qk[:, C:], vt, ks1, va1, phase=2, vfloor=va2, k8=k8, vt8=vt8, **own)
                    kw8.update(k2=k28, vt2=vt28, kscale2=ks2, vamax2=va2)
                else:
                    ops.attention_fp8_quantize(L, st, qk[:, C:], vt, ks1, va1, phase=2, k8=k8, vt8=vt8, **own)
                ops.attention_fp8(L, st, qk, k8, vt8, o, ks1, va1, n_images=n, heads=self.heads, D=Dh, Lq=N, L1=N, ldq=ldqk,
                                  ldk=C, ldvt=ldvt, ldo=C, **kw8)
            else:
                ops.attention(L, st, qk, qk[:, C:], vt, o, n_images=n, heads=self.heads, D=C // self.heads, Lq=N, L1=N,
                              ldq=ldqk, ldk=ldqk, ldvt=ldvt, ldo=C, **kw)
            cc = self.cross_const[prefix]  # [batch entries][C]: the folded 1-key cross-attention, one row per CFG half
            if self.cfg_half is not None and B == 1:
                cc = cc[self.cfg_half:self.cfg_half + 1]
            self.run.gemm_ln(o, w[t + ".attn1.to_out.0.w"], hid, bias=w[t + ".attn1.to_out.0.bias"],
                             rowvec=cc, rowvec_period=F * N, residual=hid)
            feed_forward(t + ".ff1", t + ".ff.net.2", hid)
            self.run.gemm_with_stats(hid, w[prefix + ".proj_out.w"], x, N, bias=w[prefix + ".proj_out.bias"], residual=x2d)
            return x

        def motion(prefix, x):
            """VanillaTemporalModule / TemporalTransformer3DModel.forward (motion_module.py:146-259)."""
            _, h, ww, C = x.shape
            N = h * ww
            M = n * N
            mm = prefix + ".temporal_transformer"
            x2d = x.view(M, C)
            sc, sh = gn_affine(x, mm + ".norm", 1e-6)
            hid = ws.get(f"tr_h_{M}x{C}", (M, C))
            proj_in(x2d, sc, sh, N, w[mm + ".proj_in.w"], w[mm + ".proj_in.bias"], hid)
            mmk = self.cfg["motion_module_kwargs"]
            for li in range(mmk.get("num_transformer_block", 1)):
                b = f"{mm}.transformer_blocks.{li}"
                for ai in range(len(mmk["attention_block_types"])):
                    self.run.temporal_attention_block(f"{b}.attention_blocks.{ai}", hid, B, F, N, sharded)
                feed_forward(b + ".ff1", b + ".ff.net.2", hid)
            self.run.gemm_with_stats(hid, w[mm + ".proj_out.w"], x, N, bias=w[mm + ".proj_out.bias"], residual=x2d)
            return x

        def tap(name, v):
            if self.tap is not None:
                self.tap(name, v)

        def tap_fine(name, v):  # the activations BETWEEN the tap points above (tests/test_gpu_storage_model.py: teacher-forced blocks)
            if self.tap_fine is not None:
                self.tap_fine(name, v)

        # ---- conv_in (+ pose/camera conditioning)  unet_3d.py:482-484
        x = ws.get("conv_in", (n, H, W, boc[0]))
        conv(x_in, w["conv_in.w"], x, bias=w["conv_in.bias"], residual=cond)
        tap_fine("conv_in", x)
        skips: List[torch.Tensor] = [x]
        for spec in self.unet.specs:
            p = spec.prefix
            if spec.kind == "down":
                for j in range(len(spec.resnets)):
                    x = resnet(f"{p}.resnets.{j}", x, None, f"{p}.{j}")
                    tap(f"{p}.resnets.{j}", x)
                    if spec.has_attn:
                        x = transformer(f"{p}.attentions.{j}", x)
                        tap(f"{p}.attentions.{j}", x)
                    if spec.has_motion:
                        x = motion(f"{p}.motion_modules.{j}", x)
                        tap(f"{p}.motion_modules.{j}", x)
                    skips.append(x)
                if spec.resample:
                    _, h, ww, C = x.shape
                    y = ws.get(f"{p}.down", (n, (h + 1) // 2, (ww + 1) // 2, C))
                    conv(x, w[f"{p}.downsamplers.0.conv.w"], y, mode=A.CONV_S2, bias=w[f"{p}.downsamplers.0.conv.bias"])
                    x = y
                    tap_fine(f"{p}.downsamplers.0", x)
                    skips.append(x)
            elif spec.kind == "mid":
                x = resnet("mid_block.resnets.0", x, None, "mid.0")
                tap_fine("mid_block.resnets.0", x)
                x = transformer("mid_block.attentions.0", x)
                tap_fine("mid_block.attentions.0", x)
                if spec.has_motion:
                    x = motion("mid_block.motion_modules.0", x)
                    tap_fine("mid_block.motion_modules.0", x)
                x = resnet("mid_block.resnets.1", x, None, "mid.1")
                tap("mid_block", x)
            else:
                for j in range(len(spec.resnets)):
                    x = resnet(f"{p}.resnets.{j}", x, skips.pop(), f"{p}.{j}")
                    tap_fine(f"{p}.resnets.{j}", x)
                    if spec.has_attn:
                        x = transformer(f"{p}.attentions.{j}", x)
                        tap_fine(f"{p}.attentions.{j}", x)
                    if spec.has_motion:
                        x = motion(f"{p}.motion_modules.{j}", x)
                    tap(f"{p}.{j}", x)
                if spec.resample:
                    _, h, ww, C = x.shape
                    y = ws.get(f"{p}.up", (n, 2 * h, 2 * ww, C))
                    conv(x, w[f"{p}.upsamplers.0.conv.w"], y, mode=A.CONV_UP2, bias=w[f"{p}.upsamplers.0.conv.bias"])
                    x = y
                    tap_fine(f"{p}.upsamplers.0", x)
        if self.kind == "reference":
            return x  # conv_norm_out / conv_out are removed in the ReferenceNet (unet_2d_condition.py:1295-1299)
        sc, sh = gn_affine(x, "conv_norm_out", self.eps)
        y = ws.get("conv_out", (n, H, W, self.cfg["out_channels"]))
        ops.conv3x3(L, st, x, w["conv_out.w"], y, pro_scale=sc, pro_shift=sh, pro_act=A.ACT_SILU, bias=w["conv_out.bias"])
        tap_fine("conv_out", y)
        return y
