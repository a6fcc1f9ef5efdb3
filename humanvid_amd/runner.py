"""Launch sequencing helpers shared by the native executors (UNet3D, ReferenceNet, camera encoder,
pose guider): each method is one fused block of the reference expressed as C-ABI launches.
See engine.py for the fusion map and the reference citations."""
from __future__ import annotations

from typing import Dict, Optional

import os

import torch

from . import _abi as A
from . import lib as hvlib
from . import ops

BF16 = torch.bfloat16
F32 = torch.float32


class Workspace:
    """Named persistent device buffers (stable addresses => HIP-graph friendly)."""

    def __init__(self, device):
        self.device = device
        self.bufs: Dict[str, torch.Tensor] = {}
        # tables keyed by an activation's ADDRESS (Runner.gn_parts / ln_parts: normalisation statistics left by the producer,
        # with their "gnp_<address>" / "lnp_<address>" buffers here): purged when the activation buffer is reallocated
        self.address_tables = []

    def get(self, name: str, shape, dtype=BF16) -> torch.Tensor:
        shape = tuple(int(s) for s in shape)
        n = 1
        for s in shape:
            n *= s
        t = self.bufs.get(name)
        if t is None or t.dtype != dtype or t.numel() < n:
            if t is not None:  # grown (e.g. a larger batch): what was keyed by the old address dies with it
                old = t.data_ptr()
                for pre in ("gnp_", "lnp_"):
                    self.bufs.pop(f"{pre}{old}", None)
                for table in self.address_tables:
                    table.pop(old, None)
            t = torch.empty(max(n, 1), dtype=dtype, device=self.device)
            self.bufs[name] = t
        return t[:n].view(shape)

    def bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.bufs.values())


class FrameShard:
    """Frame-axis sharding of one context window over the ranks of a torch.distributed group
    (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""

    def __init__(self, group=None, window_groups: int = 1, cfg_groups: int = 1):
        """window_groups = G > 1 (long clips with several context windows per step, e.g. config #5):
        the ranks of `group` are split into G consecutive sub-groups; sub-group g takes the context windows i with
        i % G == g and shards THEIR frames over its own ranks only (smaller all-to-all groups, no exchange between
        windows); the per-step noise accumulator is all-reduced over the whole group as before.
        cfg_groups = 2 (round 6): the CFG-parallel axis.  The two halves of a guided step -- the unconditional and the
        conditional forward -- do not interact before the guidance combine (pipeline_pose2vid_long.py:555-559): the ranks are
        split into two consecutive sub-groups, sub-group h runs the B = 1 forward of half h (Engine.clone_for_half) and
        shards ITS frames over its own ranks; each sub-group fills its half of the noise accumulator and the accumulator
        all-reduce over all ranks (which the sharded step issues anyway) hands both halves to everyone.  At two ranks the
        frame groups have ONE rank each: no temporal exchange at all, one 4.7 MB all-reduce per step."""
        import torch.distributed as dist

        self.dist = dist
        self.all_group = group  # accumulator all-reduce (every rank of the job)
        self.window_groups = int(window_groups)
        self.window_group = 0
        self.cfg_groups = int(cfg_groups)
        self.cfg_group = 0
        if self.cfg_groups not in (1, 2):
            raise ValueError("cfg_groups is 1 or 2 (a guided step has two halves)")
        if self.cfg_groups > 1 and self.window_groups > 1:
            raise NotImplementedError("cfg_groups and window_groups split the same ranks: use one of them")
        splits = max(self.window_groups, self.cfg_groups)
        if splits > 1:
            total, me = dist.get_world_size(group), dist.get_rank(group)
            if total % splits:
                raise ValueError(f"{total} ranks do not split into {splits} {'CFG' if self.cfg_groups > 1 else 'window'} groups")
            per = total // splits
            base = dist.get_process_group_ranks(group) if group is not None else list(range(total))
            subs = [dist.new_group(ranks=base[g * per:(g + 1) * per]) for g in range(splits)]  # collective
            if self.cfg_groups > 1:
                self.cfg_group = me // per
            else:
                self.window_group = me // per
            group = subs[me // per]
        self.group = group  # frame sharding + temporal exchange
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # every rank of the job (== world / rank without window groups): what is not tied to a window -- the VAE decode of
        # the finished clip -- is split over all of them
        self.all_world = dist.get_world_size(self.all_group)
        self.all_rank = dist.get_rank(self.all_group)

        # gloo (CPU tests, or several ranks sharing one GPU in the 1-GPU parity test) has no device
        # collectives: stage through host memory.  nccl (= RCCL over xGMI) runs on device buffers.
        self.staged = dist.get_backend(group) != "nccl"
        self.recorder = None  # set by the pipeline while it records a step as command-list segments
        # temporal-attention exchange: "alltoall" re-shards frames <-> pixels around the attention (each rank sends
        # 7/8 of its own q|k|v and gets its output back: 1.5 GB per rank and step at config #4), "allgather" replicates
        # K/V of all frames on every rank (6.0 GB); SURVEY.md section 8(e)
        self.exchange = os.environ.get("HUMANVID_TEMPORAL_EXCHANGE", "alltoall")
        # exchange / compute overlap of a guided (CFG) step: the two halves on two streams, replayed interleaved
        # (Pose2VideoPipeline.denoise, StepRecorder.replay_interleaved).  OPT-IN (HUMANVID_CFG_STREAMS=1, bench.py --cfg-streams 1):
        # the choreography -- collectives issued from two streams next to graph-replayed segments -- is verified bit for bit
        # on the host-staged transport and on a ONE-rank RCCL group (tests/test_gpu_sharded.py), but it has never run with more
        # than one rank on the real transport, and a rank that fails inside a collective would hang its peers: until a
        # multi-rank run has shown bit-identity against the serial replay, the serial replay is the default everywhere
        # (ADVICE round 4).
        self.overlap_cfg = os.environ.get("HUMANVID_CFG_STREAMS") == "1"
        # the recorded step -- command-list segments AND the collectives between them -- captured as ONE device graph
        # (StepRecorder.capture_step_graph): one graph launch per step instead of ~85 segment launches + ~85 torch.distributed
        # calls.  Device collectives only (RCCL calls are recorded into an open stream capture; the host-staged gloo transport
        # of the CPU / one-GPU tests cannot be).  OPT-IN (HUMANVID_STEP_GRAPH=1, bench.py --step-graph 1) for the same reason
        # as the two-stream replay: verified bit for bit against the segment replay on a ONE-rank RCCL group
        # (tests/test_gpu_sharded.py), never run with more than one rank.
        self.step_graph = os.environ.get("HUMANVID_STEP_GRAPH") == "1"
        # diagnostics for the scaling runs (bench.py --gpus N): with `measure` on, every collective is counted with the bytes
        # this rank sends and bracketed by two events on the compute stream -- the kernels behind a collective wait for
        # it, so the event distance is the exchange time the step is EXPOSED to (nothing overlaps it yet)
        self.measure = False
        self.stats = dict(collectives=0, bytes_sent=0, events=[])
        # Diagnostic: run the SHARDED code path (exchange layouts, command-list segments cut at the collectives, CFG halves on
        # two streams) with a group of ONE rank -- every collective then is a local copy through the real backend.  This is how
        # the RCCL choreography (collectives issued under two streams next to graph-replayed segments) is exercised on a
        # one-GPU box, where RCCL refuses a second rank per device (tests/test_gpu_sharded.py).  Never set by the product.
        self.single_rank_sharded = os.environ.get("HUMANVID_SINGLE_RANK_SHARDED") == "1"

    @property
    def active(self) -> bool:
        """does this job take the sharded path?  (more than one rank in the frame group, or the one-rank diagnostic)"""
        return self.world > 1 or self.single_rank_sharded

    def _timed(self, nbytes: int, fn):
        if not self.measure:
            return fn()
        self.stats["collectives"] += 1
        self.stats["bytes_sent"] += int(nbytes)
        if torch.cuda.is_available():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn()
            e1.record()
            self.stats["events"].append((e0, e1))
            return out
        return fn()

    def reset_stats(self):
        self.stats = dict(collectives=0, bytes_sent=0, events=[])

    def exposed_ms(self) -> float:
        """sum of the event distances around the measured collectives (call after a stream synchronisation)"""
        return float(sum(a.elapsed_time(b) for a, b in self.stats["events"]))

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor):
        if self.recorder is not None:
            return self.recorder.collective(lambda: self._all_gather(out, inp))
        return self._all_gather(out, inp)

    def all_reduce(self, t: torch.Tensor):
        if self.recorder is not None:
            return self.recorder.collective(lambda: self._all_reduce(t))
        return self._all_reduce(t)

    def all_to_all(self, out: torch.Tensor, inp: torch.Tensor):
        """equal-split all-to-all of flat, contiguous buffers: chunk r of `inp` goes to rank r, chunk s of `out` comes
        from rank s (runs immediately: callers wrap whole exchange closures with `deferred`)"""
        sent = inp.numel() * inp.element_size() * (self.world - 1) // max(1, self.world)
        if self.staged and inp.device.type != "cpu":
            def staged():
                host_in, host_out = inp.cpu(), torch.empty(out.numel(), dtype=inp.dtype)
                self.dist.all_to_all_single(host_out, host_in.reshape(-1), group=self.group)
                out.view(-1).copy_(host_out)
            return self._timed(sent, staged)
        self._timed(sent, lambda: self.dist.all_to_all_single(out.view(-1), inp.view(-1), group=self.group))

    def deferred(self, fn):
        """run `fn` (host-side exchange code: torch copies + a collective) now, or -- while a step is being recorded
        as command-list segments -- at this point of every replay"""
        if self.recorder is not None:
            return self.recorder.collective(fn)
        return fn()

    def _all_gather(self, out: torch.Tensor, inp: torch.Tensor, everyone: bool = False):
        group = self.all_group if everyone else self.group
        sent = inp.numel() * inp.element_size() * ((self.all_world if everyone else self.world) - 1)
        if self.staged and inp.device.type != "cpu":
            def staged():
                host_out = torch.empty(out.numel(), dtype=inp.dtype)
                self.dist.all_gather_into_tensor(host_out, inp.cpu().reshape(-1), group=group)
                out.view(-1).copy_(host_out)
            return self._timed(sent, staged)
        self._timed(sent, lambda: self.dist.all_gather_into_tensor(out, inp, group=group))

    def all_gather_everyone(self, out: torch.Tensor, inp: torch.Tensor):
        """all-gather over EVERY rank of the job (with window groups: across the groups), outside any recorded step"""
        return self._all_gather(out, inp, everyone=True)

    def _all_reduce(self, t: torch.Tensor):
        # over ALL ranks of the job: with window groups the other groups hold the other windows' contributions
        total = self.dist.get_world_size(self.all_group)
        sent = 2 * t.numel() * t.element_size() * (total - 1) // max(1, total)  # ring: reduce-scatter + all-gather
        if self.staged and t.device.type != "cpu":
            def staged():
                h = t.cpu()
                self.dist.all_reduce(h, group=self.all_group)
                t.copy_(h)
            return self._timed(sent, staged)
        self._timed(sent, lambda: self.dist.all_reduce(t, group=self.all_group))

    def frame_range(self, n_frames: int):
        if n_frames % self.world:
            raise ValueError(f"{n_frames} frames do not shard evenly over {self.world} ranks")
        per = n_frames // self.world
        return self.rank * per, per


class Runner:
    def __init__(self, device, w: Dict[str, torch.Tensor], ws: Optional[Workspace] = None, groups: int = 32,
                 shard: Optional[FrameShard] = None, temporal_heads: int = 8):
        self.lib = hvlib.load()
        self.device = device
        self.w = w
        self.ws = ws or Workspace(device)
        self.groups = groups
        self.shard = shard
        self.temporal_heads = int(temporal_heads)  # num_attention_heads / temporal_attention_nhead of the config
        if self.temporal_heads != 8:
            raise NotImplementedError(f"hv_temporal_attention is built for 8 heads, the config asks for {self.temporal_heads}")
        self._pe_split: Dict[tuple, tuple] = {}
        self.gn_parts: Dict[int, torch.Tensor] = {}
        self.ln_parts: Dict[int, torch.Tensor] = {}  # the same for LayerNorm row statistics (gemm_ln / ln_stats)
        # (a cached workspace that is re-linked to a fresh Runner -- Engine.clone_for_half per denoise() call -- keeps only
        #  the LIVE runner's tables: the previous clone's are unreachable, and Workspace.get walks this list on every regrowth)
        self.ws.address_tables = [self.gn_parts, self.ln_parts]
        # GroupNorm / LayerNorm statistics come from the producers' epilogues (gn_part / ln_part) wherever the producing
        # kernel can leave them; GroupNorm apply + SiLU in front of a ResnetBlock3D convolution (resnet.py:215-222, 235-241)
        # is its own pass (hv_affine_apply / _cat into a scratch activation): the convolution's operand prologue repeats the
        # transform in every 128-channel output tile (3 .. 10 of them) and halo pixel (x 1.4) on the k-loop's critical path --
        # 0.83 vs 0.64 ms per 320 -> 320 convolution at 96 x 64 x 48 images against 0.08 ms for the pass
        # (profiles/r03_conv_apply_ab.txt; the A/B environment switches of rounds 2-4 are retired).

    @property
    def st(self) -> int:
        return hvlib.current_stream()

    # ---- normalisation statistics ---------------------------------------------------------------
    # GroupNorm partial statistics left by the kernel that PRODUCED an activation (hv_conv3x3 / hv_gemm gn_part), keyed by
    # the activation's address: gn_affine then needs no pass over the activation.  An entry is valid for the buffer's
    # current contents only: every producer either refreshes it (conv_with_stats / gemm_with_stats) or drops it.
    def conv_with_stats(self, x, wt, y, **kw):
        """ops.conv3x3 that also leaves the GroupNorm partial statistics of y"""
        if kw.get("pro_scale") is not None and wt.shape[0] > 128:
            n, h, ww, c1 = x.shape
            x2 = kw.pop("x2", None)
            ctot = c1 + (0 if x2 is None else x2.shape[3])
            xn = self.ws.get(f"conv_xn_{n}x{h}x{ww}x{ctot}", (n, h, ww, ctot))
            ops.affine_apply(self.lib, self.st, x.view(-1, c1), kw.pop("pro_scale"), kw.pop("pro_shift"), xn.view(-1, ctot),
                             rows_per_image=h * ww, act=kw.pop("pro_act", A.ACT_NONE),
                             x2=None if x2 is None else x2.view(-1, x2.shape[3]))
            x = xn
        parts = ops.conv3x3(self.lib, self.st, x, wt, y, query_gn_parts=True, **kw)
        if parts <= 0:
            self.gn_parts.pop(y.data_ptr(), None)
            ops.conv3x3(self.lib, self.st, x, wt, y, **kw)
            return
        part = self.ws.get(f"gnp_{y.data_ptr()}", (y.shape[0], parts, y.shape[3], 2), F32)
        ops.conv3x3(self.lib, self.st, x, wt, y, gn_part=part, **kw)
        self.gn_parts[y.data_ptr()] = part

    def gemm_with_stats(self, x2d, wt, y4d, rows_per_image, **kw):
        """ops.gemm writing the [n, h, w, C] activation y4d (as rows) that also leaves its GroupNorm partial statistics
        where the problem allows (hv_gemm_gn_parts), and drops stale ones where it does not"""
        y2d = y4d.view(-1, y4d.shape[3])
        parts = ops.gemm(self.lib, self.st, x2d, wt, y2d, gn_rows_per_image=rows_per_image, query_gn_parts=True, **kw)
        if parts <= 0:
            self.gn_parts.pop(y4d.data_ptr(), None)
            ops.gemm(self.lib, self.st, x2d, wt, y2d, **kw)
            return
        part = self.ws.get(f"gnp_{y4d.data_ptr()}", (y4d.shape[0], parts, y4d.shape[3], 2), F32)
        ops.gemm(self.lib, self.st, x2d, wt, y2d, gn_part=part, gn_rows_per_image=rows_per_image, **kw)
        self.gn_parts[y4d.data_ptr()] = part

    def gn_affine(self, x, prefix, eps, x2=None):
        n = x.shape[0]
        C = x.shape[3] + (0 if x2 is None else x2.shape[3])
        sc = self.ws.get("gn_scale", (n, C), F32)
        sh = self.ws.get("gn_shift", (n, C), F32)
        pixels = x.shape[1] * x.shape[2]
        p1 = self.gn_parts.get(x.data_ptr())
        p2 = None if x2 is None else self.gn_parts.get(x2.data_ptr())
        if p1 is not None and (x2 is None or p2 is not None):
            ops.groupnorm_from_parts(self.lib, self.st, p1, self.w[prefix + ".g"], self.w[prefix + ".b"], self.groups, eps,
                                     pixels, sc, sh, part2=p2)
            return sc, sh
        splits = max(1, min(64, pixels // 48))
        part = self.ws.get("gn_partial", (n * 64 * self.groups * 2,), F32)
        ops.groupnorm_affine(self.lib, self.st, x, self.w[prefix + ".g"], self.w[prefix + ".b"], self.groups, eps, part,
                             sc, sh, x2=x2, splits=splits)
        return sc, sh

    def ln_stats(self, h2d):
        M = h2d.shape[0]
        mean, rstd = self.ws.get("ln_mean", (M,), F32), self.ws.get("ln_rstd", (M,), F32)
        part = self.ln_parts.get(h2d.data_ptr())
        if part is not None and part.shape[0] == M:
            ops.layernorm_from_parts(self.lib, self.st, part, h2d.shape[1], mean, rstd)
        else:
            ops.layernorm_stats(self.lib, self.st, h2d, mean, rstd)
        return mean, rstd

    def gemm_ln(self, x2d, wt, y2d, **kw):
        """ops.gemm whose output feeds a LayerNorm: leaves the row sums of what it stores (hv_gemm ln_part) where the problem
        allows, so that ln_stats(y2d) needs no pass over the activation; drops stale ones where it does not"""
        parts = ops.gemm(self.lib, self.st, x2d, wt, y2d, query_ln_parts=True, **kw)
        if parts <= 0:
            self.ln_parts.pop(y2d.data_ptr(), None)
            ops.gemm(self.lib, self.st, x2d, wt, y2d, **kw)
            return
        part = self.ws.get(f"lnp_{y2d.data_ptr()}", (y2d.shape[0], parts, 2), F32)
        ops.gemm(self.lib, self.st, x2d, wt, y2d, ln_part=part, **kw)
        self.ln_parts[y2d.data_ptr()] = part

    # ---- LN -> GEGLU feed-forward -> +residual (in place) ----------------------------------------
    def feed_forward(self, ff1, ff2, h2d):
        w = self.w
        M, C = h2d.shape
        mean, rstd = self.ln_stats(h2d)
        ffh = self.ws.get(f"ffh_{M}x{C}", (M, 4 * C))
        ops.gemm(self.lib, self.st, h2d, w[ff1 + ".w"], ffh, bias=w[ff1 + ".bias"], row_mean=mean, row_rstd=rstd,
                 colsum=w[ff1 + ".colsum"], geglu=True)
        ops.gemm(self.lib, self.st, ffh, w[ff2 + ".w"], h2d, bias=w[ff2 + ".bias"], residual=h2d)
        self.ln_parts.pop(h2d.data_ptr(), None)  # rewritten without statistics (no LayerNorm reads it next)

    # ---- LN(+PE) -> qkv -> attention over frames -> out-proj + residual (in place) -----------------
    def temporal_attention_block(self, ab, hid, B, F, N, sharded: bool):
        """`ab` = weight prefix of one VersatileAttention / TemporalSelfAttention block; hid [(B F N), C]."""
        w, L, st, ws = self.w, self.lib, self.st, self.ws
        M, C = hid.shape
        H = self.temporal_heads
        D = C // H
        mean, rstd = self.ln_stats(hid)
        pe = w.get(ab + ".qkv.pe")
        f_total = F * (self.shard.world if sharded else 1)
        if pe is not None and f_total > pe.shape[0]:
            # the reference fails here too: x + pe[:, :x.size(1)] does not broadcast (motion_module.py:262-277)
            raise ValueError(f"{f_total} frames per window exceed temporal_position_encoding_max_len = {pe.shape[0]} ({ab})")
        o = ws.get(f"tr_o_{M}x{C}", (M, C))
        if not sharded:
            pekw = {} if pe is None else dict(pe=pe, pe_period=N, pe_frames=F)
            qkv = ws.get(f"mm_qkv_{M}x{C}", (M, 3 * C))
            ops.gemm(L, st, hid, w[ab + ".qkv.w"], qkv, bias=w[ab + ".qkv.bias"], row_mean=mean, row_rstd=rstd,
                     colsum=w[ab + ".qkv.colsum"], **pekw)
            ops.temporal_attention(L, st, qkv, o, B=B, F=F, P=N, heads=H, D=D)
        elif self.shard.exchange == "alltoall" and N % self.shard.world == 0:
            # frames <-> pixels re-sharding: project q|k|v for the local frames, all-to-all so that this rank holds ALL
            # frames of its 1/R slice of the pixels, attend locally with the unsharded kernel, all-to-all the result back
            R, f0 = self.shard.world, self.shard.rank * F
            Np = N // R
            pekw = {}
            if pe is not None:
                key = (ab, f0, F, "full")
                if key not in self._pe_split:
                    self._pe_split[key] = pe[f0:f0 + F].contiguous()
                pekw = dict(pe=self._pe_split[key], pe_period=N, pe_frames=F)
            # No re-ordering copies: the QKV projection stores its rows straight in the send layout [dest rank r][b][f][p]
            # (hv_gemm row permutation), the temporal kernel addresses the received [src rank s][b][f][p] chunks in place and
            # writes its output in the same order (= the send buffer of the way back), and the output projection below folds
            # the inverse re-ordering into its row permutation.
            send = ws.get(f"mm_a2a_s_{M}x{C}", (R, B, F, Np, 3 * C))
            recv = ws.get(f"mm_a2a_r_{M}x{C}", (R, B, F, Np, 3 * C))
            ops.gemm(L, st, hid, w[ab + ".qkv.w"], send.view(M, 3 * C), bias=w[ab + ".qkv.bias"], row_mean=mean, row_rstd=rstd,
                     colsum=w[ab + ".qkv.colsum"], row_perm=(B * F, R, Np), **pekw)
            shard = self.shard
            shard.deferred(lambda: shard.all_to_all(recv, send))                      # chunk s = frames of rank s, my pixels
            send_o = send.view(-1)[:M * C].view(R, B, F, Np, C)
            recv_o = recv.view(-1)[M * C:2 * M * C].view(R, B, F, Np, C)             # disjoint from the rows still being read
            ops.temporal_attention_exchanged(L, st, recv, send_o, B=B, F_local=F, ranks=R, P=Np, heads=H, D=D)
            shard.deferred(lambda: shard.all_to_all(recv_o, send_o))                  # chunk r = pixel slice r, my frames
            self.gemm_ln(recv_o.view(M, C), w[ab + ".to_out.0.w"], hid, bias=w[ab + ".to_out.0.bias"], residual=hid,
                         row_perm=(R, B * F, Np))
            return
        else:
            R, f0 = self.shard.world, self.shard.rank * F
            pq, pkv = {}, {}
            if pe is not None:
                key = (ab, f0, F)
                if key not in self._pe_split:
                    self._pe_split[key] = (pe[f0:f0 + F, :C].contiguous(), pe[f0:f0 + F, C:].contiguous())
                a, b = self._pe_split[key]
                pq, pkv = dict(pe=a, pe_period=N, pe_frames=F), dict(pe=b, pe_period=N, pe_frames=F)
            q = ws.get(f"mm_q_{M}x{C}", (M, C))
            kvl = ws.get(f"mm_kv_{M}x{C}", (M, 2 * C))
            wq, bq, cs = w[ab + ".qkv.w"], w[ab + ".qkv.bias"], w[ab + ".qkv.colsum"]
            ops.gemm(L, st, hid, wq[:C], q, bias=bq[:C], row_mean=mean, row_rstd=rstd, colsum=cs[:C], **pq)
            ops.gemm(L, st, hid, wq[C:], kvl, bias=bq[C:], row_mean=mean, row_rstd=rstd, colsum=cs[C:], **pkv)
            kvg = ws.get(f"mm_kvg_{M}x{C}", (R, B, F, N, 2 * C))
            self.shard.all_gather(kvg.view(-1), kvl.view(-1))
            ops.temporal_attention_sharded(L, st, q, kvg, o, B=B, Fq=F, ranks=R, P=N, heads=H, D=D)
        self.gemm_ln(o, w[ab + ".to_out.0.w"], hid, bias=w[ab + ".to_out.0.bias"], residual=hid)

