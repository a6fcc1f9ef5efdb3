"""Pose2VideoPipeline -- drop-in for /root/reference/src/pipelines/pipeline_pose2vid_long.py:35-588.

`__call__` keeps the reference's signature and semantics (CLIP embed, ReferenceNet write pass at
i == 0, context windows, CFG, DDIM v-prediction, per-frame VAE decode).  The denoising loop body
(:454-571) is `denoise()`: everything timestep-independent is hoisted out of the loop (PoseGuider,
CameraPoseEncoder, window lists, folded cross-attention constants, reference-bank K/V), one step is
a fixed launch sequence into libhumanvid_hip.so (captured as a HIP graph and replayed), and with a
`FrameShard` the frames of each window are split over the ranks of a node (frames <-> pixels all-to-all
around every temporal attention -- or an all-gather of its K/V --, all-reduce of the tiny noise
accumulator per step; the step is then replayed as command-list segments cut at the collectives).

VAE and CLIP stay stock PyTorch-ROCm modules supplied by the caller (BASELINE.json north_star).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Union

import numpy as np
import torch

from . import lib as hvlib
from . import ops
from .reference_control import ReferenceAttentionControl
from .runner import FrameShard
from .scheduler import fused_step_coefficients, get_context_scheduler

F32 = torch.float32
BF16 = torch.bfloat16


@dataclass
class Pose2VideoPipelineOutput:
    videos: Union[torch.Tensor, np.ndarray]


def _pil_to_tensor(images, height, width, normalize: bool) -> torch.Tensor:
    """VaeImageProcessor.preprocess: resize (Lanczos) to (height, width) rounded down to a multiple
    of 8, [0,1], optionally 2x-1.  -> [n,3,H,W] fp32."""
    from PIL import Image

    if not isinstance(images, (list, tuple)):
        images = [images]
    height, width = height - height % 8, width - width % 8
    out = []
    for im in images:
        im = im.convert("RGB").resize((width, height), resample=Image.LANCZOS)
        a = torch.from_numpy(np.asarray(im, dtype=np.float32) / 255.0).permute(2, 0, 1)
        out.append(2.0 * a - 1.0 if normalize else a)
    return torch.stack(out)


def _clip_preprocess(ref_image) -> torch.Tensor:
    """CLIPImageProcessor() defaults on an image already resized to 224x224 (pipeline :380-382)."""
    from PIL import Image

    im = ref_image.convert("RGB").resize((224, 224), resample=Image.BICUBIC)
    a = torch.from_numpy(np.asarray(im, dtype=np.float32) / 255.0).permute(2, 0, 1)
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(3, 1, 1)
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(3, 1, 1)
    return ((a - mean) / std)[None]


class StepRecorder:
    """Records one denoising step as native command-list segments separated by collectives
    (hv_cmdlist_*): replaying a step is then one ctypes call per segment plus the collectives,
    instead of ~1000 Python-built launches -- the frame-sharded counterpart of the HIP-graph replay."""

    def __init__(self, lib):
        self.lib = lib
        self.items = []  # ("k", list handle) | ("c", callable)
        self.graph = None  # capture_step_graph: the whole replay as ONE device graph

    def begin(self):
        self.lib.call("hv_cmdlist_begin")

    def _close(self, fn_name):
        import ctypes

        h = ctypes.c_void_p()
        self.lib.call(fn_name, ctypes.byref(h))
        if self.lib.cdll.hv_cmdlist_size(h) > 0:
            self.items.append(("k", h))
        else:
            self.lib.call("hv_cmdlist_destroy", h)

    def collective(self, fn):
        self._close("hv_cmdlist_cut")
        self.items.append(("c", fn))
        return fn()

    def end(self):
        self._close("hv_cmdlist_end")

    def replay(self):
        st = hvlib.current_stream()
        for kind, x in self.items:
            if kind == "k":
                self.lib.call("hv_cmdlist_run", x, st)
            else:
                x()

    @staticmethod
    def replay_interleaved(recorders, streams):
        """Replay several recordings of the SAME program (the two CFG halves of a step) item by item, recording r on
        stream r: segment k of every half is issued before collective k of any, so the collectives reach the communicator's
        queue in the order A1, B1, A2, B2, ... on every rank, and while half A waits for its exchange the GPU runs half B's
        segment (and the other way round one item later)."""
        n = len(recorders[0].items)
        assert all(len(r.items) == n and [k for k, _ in r.items] == [k for k, _ in recorders[0].items] for r in recorders)
        for i in range(n):
            for r, s in zip(recorders, streams):
                kind, x = r.items[i]
                with torch.cuda.stream(s):
                    if kind == "k":
                        r.lib.call("hv_cmdlist_run", x, s.cuda_stream)
                    else:
                        x()

    @staticmethod
    def capture_step_graph(lib, replay_fn):
        """Capture one whole replay -- `replay_fn` = the segments re-issued launch by launch AND the collectives between them
        (device collectives only: RCCL enqueues on the stream it is called under, and PyTorch's ProcessGroupNCCL records that
        into an open stream capture) -- as ONE torch.cuda.CUDAGraph: a step is then a single graph launch instead of one launch
        per segment plus one torch.distributed call per collective (VERDICT r4 #4b).  The capture itself executes nothing.
        The segments' own per-segment graphs are bypassed while capturing (tuning key 7: hv_cmdlist_run re-issues the recorded
        closures on the capturing stream)."""
        g = torch.cuda.CUDAGraph()
        keep = "7=0" not in os.environ.get("HUMANVID_TUNING", "").replace(" ", "").split(",")
        torch.cuda.synchronize()
        lib.call("hv_set_tuning", 7, 0)
        try:
            # thread_local: ProcessGroupNCCL's watchdog thread may query events while this thread captures
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                replay_fn()
        finally:
            if keep:
                lib.call("hv_set_tuning", 7, 1)
        return g

    def destroy(self):
        self.graph = None
        for kind, x in self.items:
            if kind == "k":
                self.lib.call("hv_cmdlist_destroy", x)
        self.items = []


class Pose2VideoPipeline:
    def __init__(self, vae, image_encoder, reference_unet, denoising_unet, pose_guider, camera_pose_encoder,
                 scheduler, image_proj_model=None, tokenizer=None, text_encoder=None):
        self.vae, self.image_encoder = vae, image_encoder
        self.reference_unet, self.denoising_unet = reference_unet, denoising_unet
        self.pose_guider, self.camera_pose_encoder = pose_guider, camera_pose_encoder
        self.scheduler = scheduler
        self.image_proj_model, self.tokenizer, self.text_encoder = image_proj_model, tokenizer, text_encoder
        self.vae_scale_factor = 8
        if vae is not None and hasattr(vae, "config") and hasattr(vae.config, "block_out_channels"):
            self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1)
        self.shard: Optional[FrameShard] = None
        self._graph = None
        self._graph_key = None

    # ---- DiffusionPipeline surface ---------------------------------------------------------------
    def to(self, device=None, dtype=None):
        for m in (self.vae, self.image_encoder, self.reference_unet, self.denoising_unet, self.pose_guider,
                  self.camera_pose_encoder):
            if m is not None and hasattr(m, "to"):
                m.to(device=device, dtype=dtype) if dtype is not None else m.to(device=device)
        return self

    def enable_vae_slicing(self):  # pipeline_pose2vid_long.py:83-87: plain delegation to the caller's VAE
        self.vae.enable_slicing()

    def disable_vae_slicing(self):
        self.vae.disable_slicing()

    def enable_sequential_cpu_offload(self, gpu_id=0):
        """pipeline_pose2vid_long.py:89-99 parks the modules in host memory between uses; this path keeps the weights, the
        reference banks and the step's HIP graph resident in the 288 GB of HBM by design -- refused rather than ignored."""
        raise NotImplementedError("sequential CPU offload contradicts the resident-weights design of the MI355X path")

    @property
    def device(self) -> torch.device:
        for m in (self.denoising_unet, self.reference_unet, self.vae):
            if m is not None and hasattr(m, "parameters"):
                for prm in m.parameters():
                    return prm.device
        return torch.device("cpu")

    @property
    def _execution_device(self) -> torch.device:  # pipeline_pose2vid_long.py:101-112 (no accelerate hooks here)
        return self.device

    def enable_frame_sharding(self, group=None, window_groups: Optional[int] = None, cfg_groups: Optional[int] = None):
        """Shard every context window along the frame axis over the ranks of `group` (one process per
        GPU, torch.distributed backend "nccl" = RCCL over xGMI).  window_groups = G (default: HUMANVID_WINDOW_GROUPS or 1)
        splits the ranks into G sub-groups that take different context windows of a step (window-parallel x frame-shard,
        for clips with several windows per step); cfg_groups = 2 (default: HUMANVID_CFG_GROUPS or 1) splits them into two
        sub-groups that take the two CFG halves of a guided step (CFG-parallel x frame-shard) -- see FrameShard."""
        if window_groups is None:
            window_groups = int(os.environ.get("HUMANVID_WINDOW_GROUPS", "1"))
        if cfg_groups is None:
            cfg_groups = int(os.environ.get("HUMANVID_CFG_GROUPS", "1"))
        self.shard = FrameShard(group, window_groups=window_groups, cfg_groups=cfg_groups)
        if self.denoising_unet is not None:
            self.denoising_unet._engine = None  # rebuilt with the shard on first use
        return self

    def progress_bar(self, iterable=None, total=None):
        try:
            from tqdm.auto import tqdm

            return tqdm(iterable, total=total)
        except Exception:  # pragma: no cover
            class _Null:
                def __enter__(s): return s
                def __exit__(s, *a): return False
                def update(s, n=1): pass
            return _Null()

    def prepare_latents(self, batch_size, num_channels_latents, width, height, video_length, dtype, device, generator,
                        latents=None):
        shape = (batch_size, num_channels_latents, video_length, height // self.vae_scale_factor,
                 width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(
                f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            gdev = generator.device if isinstance(generator, torch.Generator) else torch.device("cpu")
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=F32).to(device)  # randn_tensor semantics
        else:
            latents = latents.to(device)
        return latents * self.scheduler.init_noise_sigma

    @torch.no_grad()
    def decode_latents(self, latents, frames_per_batch: int = 8):
        """pipeline_pose2vid_long.py:114-127 decodes one frame per `vae.decode` call, serially, and copies each
        result to the host.  Same arithmetic here (the VAE stays on PyTorch-ROCm and treats frames as independent
        batch items), scheduled differently: `frames_per_batch` frames per call,
        one host copy at the end, and -- when frame sharding is on -- every rank decodes only its own frames and
        the pixels are all-gathered."""
        video_length = latents.shape[2]
        latents = 1 / 0.18215 * latents
        b = latents.shape[0]
        flat = latents.permute(0, 2, 1, 3, 4).reshape(b * video_length, *latents.shape[1:2], *latents.shape[3:])
        # over every rank of the job: with window groups (FrameShard(window_groups=G)) the clip is finished and replicated on
        # all of them, so the decode splits over all G x R ranks, not over the sub-group's R
        world = 1 if self.shard is None else self.shard.all_world
        lo, n = 0, flat.shape[0]
        if world > 1 and flat.shape[0] % world == 0:
            n = flat.shape[0] // world
            lo = self.shard.all_rank * n
        step = max(1, int(frames_per_batch))
        frames = [self.vae.decode(flat[i:min(i + step, lo + n)].to(self.vae.dtype)).sample for i in range(lo, lo + n, step)]
        video = torch.cat(frames)
        if n != flat.shape[0]:
            full = torch.empty(flat.shape[0], *video.shape[1:], dtype=video.dtype, device=video.device)
            self.shard.all_gather_everyone(full, video.contiguous())
            video = full
        video = video.view(b, video_length, *video.shape[1:]).permute(0, 2, 1, 3, 4)
        video = (video / 2 + 0.5).clamp(0, 1)
        return video.cpu().float().numpy()

    # ---- the hot path ----------------------------------------------------------------------------
    def build_conditioning(self, pose_cond_tensor, camera_embedding, windows: List[List[int]], f0: int, fl: int):
        """Timestep-independent conditioning per window: PoseGuider + CameraPoseEncoder feature
        (pipeline :526-539, hoisted).  Returns a list of [f_local, h, w, 320] bf16 tensors.
        `camera_embedding` is the reference's Pluecker map [1,6,F,H,W], or a (K [F,4], c2w [F,4,4]) pair of camera
        parameters (humanvid_amd.camera.cameras_to_params) for the on-device front-end, which never builds the map."""
        out = []
        for c in windows:
            pose_w = pose_cond_tensor[:, :, c]
            pose_fea = self.pose_guider.forward_nhwc(pose_w)  # [(1 f), h, w, 320]
            if camera_embedding is not None and self.camera_pose_encoder is None:
                raise ValueError("a camera embedding was passed to a pipeline built without camera_pose_encoder: it would be "
                                 "dropped silently (pass camera_embedding=None for the Animate-Anyone mode)")
            if camera_embedding is None:
                feat = pose_fea  # pipeline_pose2vid.py (Animate-Anyone mode): pose feature only
            elif isinstance(camera_embedding, (tuple, list)):
                K, c2w = camera_embedding
                idx = torch.as_tensor(c, dtype=torch.long)
                feat = self.camera_pose_encoder.forward_nhwc_from_cameras(K[idx], c2w[idx], pose_w.shape[-2], pose_w.shape[-1],
                                                                          add=pose_fea)
            else:
                feat = self.camera_pose_encoder.forward_nhwc(camera_embedding[:, :, c].float(), add=pose_fea)
            out.append(feat[f0:f0 + fl].clone())
        return out

    @torch.no_grad()
    def denoise(self, latents: torch.Tensor, pose_cond_tensor: torch.Tensor, camera_embedding: torch.Tensor,
                clip_image_embeds: torch.Tensor, num_inference_steps: int, guidance_scale: float,
                context_schedule="uniform", context_frames=24, context_stride=1, context_overlap=4,
                use_graph: bool = True, callback: Optional[Callable] = None, callback_steps: int = 1,
                max_steps: Optional[int] = None, step_hook: Optional[Callable] = None,
                after_loop: Optional[Callable] = None, first_step: int = 0) -> torch.Tensor:
        """Denoising loop body of the reference (pipeline_pose2vid_long.py:454-571).

        latents [1,4,F,h,w] fp32 (cuda); pose_cond_tensor [1,3,F,H,W] in [0,1]; camera_embedding
        [1,6,F,H,W]; clip_image_embeds [1,768] or [1,1,768].  The reference banks must already be on
        the denoising UNet (ReferenceAttentionControl.update, or engine.set_reference_banks).
        step_hook(i) runs after step i; after_loop(one_step) receives the step closure once the loop is done.
        first_step > 0 enters the schedule at that step with `latents` as they are (partial trajectories: the parity test of
        the schedule's last step, tests/test_gpu_fullsize_steps.py; the reference always starts at 0).
        callback: called as the reference calls it (pipeline_pose2vid_long.py:507, 565-571) -- its inner
        `for i in range(num_context_batches)` rebinds the step index before the callback test, so the callback receives
        (num_context_batches - 1, t, latents) with t the scheduler's 0-d timestep tensor, whenever
        (num_context_batches - 1) % callback_steps == 0 (context_batch_size = 1: one batch per context window)."""
        dev = hvlib.require_gpu()
        L = hvlib.load()
        unet = self.denoising_unet
        if self.shard is not None and unet._engine is None:
            from .engine import UNet3DEngine

            unet._engine = UNet3DEngine(unet, shard=self.shard)
        eng = unet.engine()
        do_cfg = guidance_scale > 1.0
        rep = 2 if do_cfg else 1
        sched = self.scheduler
        sched.set_timesteps(num_inference_steps)
        timesteps = [int(t) for t in sched.timesteps.tolist()]
        # hv_cfg_ddim_step fuses CFG with the DDIM update (eta = 0); its four coefficients cover v- and epsilon-prediction
        # and are derived from the scheduler's alphas_cumprod -- this module's DDIMScheduler or a diffusers-style one
        # (raises NotImplementedError for anything the fused step would follow wrongly)
        step_coeffs = [fused_step_coefficients(sched, t, num_inference_steps) for t in timesteps]
        latents = latents.to(device=dev, dtype=F32).contiguous()
        if latents.ndim != 5 or latents.shape[0] != 1:
            # pack / accumulate / cfg_ddim size their buffers for one clip (rep = CFG halves only)
            raise NotImplementedError(f"denoise() handles one clip per call (latents [1,C,F,h,w]), got {tuple(latents.shape)}")
        _, C, F_, h, w = latents.shape
        e = clip_image_embeds.reshape(1, 1, -1).to(dev)
        ehs = torch.cat([torch.zeros_like(e), e], dim=0) if do_cfg else e
        eng._banks_from_modules()
        eng.set_encoder_hidden_states(ehs)

        windows = list(get_context_scheduler(context_schedule)(0, num_inference_steps, F_, context_frames,
                                                                context_stride, context_overlap))
        world = 1 if self.shard is None else self.shard.world
        rank = 0 if self.shard is None else self.shard.rank
        if self.shard is not None and self.shard.window_groups > 1:
            # window-parallel: this rank's sub-group takes every G-th window of the step; the others contribute theirs
            # through the accumulator all-reduce
            G = self.shard.window_groups
            if G > len(windows):
                raise ValueError(f"{G} window groups for {len(windows)} context windows per step: "
                                 f"{G - len(windows)} sub-group(s) would only take part in the all-reduce")
            if len(windows) % G:
                import warnings

                warnings.warn(f"{len(windows)} context windows per step do not split evenly over {G} window groups: "
                              f"the step lasts as long as the group with {-(-len(windows) // G)} windows", stacklevel=2)
            windows = [c for i, c in enumerate(windows) if i % self.shard.window_groups == self.shard.window_group]
        plans = []  # per window: (frame index tensor of this rank, local frame count)
        for c in windows:
            if len(c) % world:
                raise ValueError(f"window of {len(c)} frames does not shard over {world} ranks")
            fl = len(c) // world
            mine = c[rank * fl:(rank + 1) * fl]
            plans.append((torch.tensor(mine, dtype=torch.int32, device=dev), fl, rank * fl))
        conds = []
        for c, (_, fl, f0) in zip(windows, plans):
            cam = camera_embedding if camera_embedding is None or isinstance(camera_embedding, (tuple, list)) \
                else camera_embedding.to(dev)
            conds.extend(self.build_conditioning(pose_cond_tensor.to(dev), cam, [c], f0, fl))

        # noise accumulator and per-frame window counter in ONE buffer: the sharded step all-reduces both with one collective
        acc_cnt = torch.zeros(rep * C * F_ * h * w + F_, dtype=F32, device=dev)
        acc = acc_cnt[:rep * C * F_ * h * w].view(rep, C, F_, h, w)
        counter = acc_cnt[rep * C * F_ * h * w:]
        t_dev = torch.zeros(rep, dtype=F32, device=dev)
        coeffs = torch.zeros(5, dtype=F32, device=dev)
        t_table = torch.tensor(timesteps, dtype=F32, device=dev)
        c_table = torch.tensor([[guidance_scale, *c] for c in step_coeffs], dtype=F32, device=dev)
        x_in = [eng.ws.get(f"pipe_x_in_{i}", (rep * fl, h, w, 32)) for i, (_, fl, _) in enumerate(plans)]
        for x in x_in:
            x.zero_()
        # ---- CFG-parallel axis (FrameShard.cfg_groups == 2): this rank's sub-group runs ONE half of the guided step as a
        # B = 1 forward and fills that half of the accumulator; the all-reduce over all ranks brings the other half
        cfg_split = self.shard is not None and self.shard.cfg_groups == 2 and do_cfg
        if cfg_split:
            hf_mine = self.shard.cfg_group
            eng_half = eng.clone_for_half(hf_mine)
            x_half_mine = [eng_half.ws.get(f"pipe_x_in_{i}", (fl, h, w, 32)) for i, (_, fl, _) in enumerate(plans)]
            for x in x_half_mine:
                x.zero_()
            counter_other = torch.zeros_like(counter)  # the window counter is accumulated once per frame: by half 0's ranks

        # ---- exchange / compute overlap for frame-sharded guided steps (FrameShard.overlap_cfg): the two CFG halves never
        # interact before hv_cfg_ddim_step, so each runs as its own B = 1 forward on its own stream (own workspace), recorded
        # once as command-list segments cut at its collectives and replayed INTERLEAVED with the other half
        sharded = self.shard is not None and self.shard.active
        overlap = sharded and do_cfg and use_graph and getattr(self.shard, "overlap_cfg", False) and self.shard.cfg_groups == 1
        if overlap:
            # the two half-engines' workspaces and their streams live on the pipeline: torch caches freed blocks per stream and
            # torch.cuda.Stream() rotates through a pool, so per-call clones re-allocated every buffer; the
            # clones themselves are re-linked per call (they share this call's banks / folded constants by reference)
            cache = self.__dict__.setdefault("_half_state", {})
            if cache.get("engine") is not eng:
                from .engine import Workspace

                cache.clear()
                cache.update(engine=eng, ws=[Workspace(eng.device), Workspace(eng.device)],
                             streams=[torch.cuda.Stream(), torch.cuda.Stream()])
            halves = [eng.clone_for_half(hf, ws=cache["ws"][hf]) for hf in (0, 1)]
            streams = cache["streams"]
            x_half = [[e.ws.get(f"pipe_x_in_{i}", (fl, h, w, 32)) for i, (_, fl, _) in enumerate(plans)] for e in halves]
            for xs in x_half:
                for x in xs:
                    x.zero_()
            counter_spare = torch.zeros_like(counter)  # the window counter is accumulated once (by half 0)

            def half_step(hf):
                st = hvlib.current_stream()
                for (frames, fl, _), cond, xi in zip(plans, conds, x_half[hf]):
                    ops.pack_ncfhw(L, st, latents, xi, rep=1, frames=frames)
                    y = halves[hf].forward_nhwc(xi, t_dev[hf:hf + 1], cond, B=1, F=fl)
                    ops.accumulate_window(L, st, y, 1, C, frames, acc[hf:hf + 1], counter if hf == 0 else counter_spare)

            def fork():
                for s_ in streams:
                    s_.wait_stream(torch.cuda.current_stream())

            def join_and_finish():
                for s_ in streams:
                    torch.cuda.current_stream().wait_stream(s_)
                self.shard.all_reduce(acc_cnt)
                ops.cfg_ddim_step(L, hvlib.current_stream(), latents, acc, counter, rep, coeffs)

        def one_step():
            if cfg_split:
                st = hvlib.current_stream()
                for (frames, fl, _), cond, xi in zip(plans, conds, x_half_mine):
                    ops.pack_ncfhw(L, st, latents, xi, rep=1, frames=frames)
                    y = eng_half.forward_nhwc(xi, t_dev[hf_mine:hf_mine + 1], cond, B=1, F=fl)
                    ops.accumulate_window(L, st, y, 1, C, frames, acc[hf_mine:hf_mine + 1], counter if hf_mine == 0 else counter_other)
                    if getattr(self.shard, "cfg_half_diagnostic", False):
                        # bench.py --cfg-half (ONE rank of a two-rank job, alone): nobody fills the other half -- stand in this
                        # half's prediction for it, so that the guided update stays a sane trajectory (with a zero half the
                        # latents blow up after step 0 and the attention kernels take their overflow re-run on every launch)
                        o = 1 - hf_mine
                        ops.accumulate_window(L, st, y, 1, C, frames, acc[o:o + 1], counter if o == 0 else counter_other)
                self.shard.all_reduce(acc_cnt)
                ops.cfg_ddim_step(L, st, latents, acc, counter, rep, coeffs)
                return
            if overlap:  # (eager form: the halves one after the other on their streams -- steps 0 and the recorded one)
                fork()
                for hf in (0, 1):
                    with torch.cuda.stream(streams[hf]):
                        half_step(hf)
                join_and_finish()
                return
            st = hvlib.current_stream()
            for (frames, fl, _), cond, xi in zip(plans, conds, x_in):
                ops.pack_ncfhw(L, st, latents, xi, rep=rep, frames=frames)
                y = eng.forward_nhwc(xi, t_dev, cond, B=rep, F=fl)
                ops.accumulate_window(L, st, y, rep, C, frames, acc, counter)
            if sharded or (self.shard is not None and self.shard.window_groups > 1):
                self.shard.all_reduce(acc_cnt)
            ops.cfg_ddim_step(L, st, latents, acc, counter, rep, coeffs)

        graph = None
        recorder = None
        # the recorded step (segments + device collectives) as ONE graph: FrameShard.step_graph, opt-in, RCCL only
        step_graph = bool(self.shard is not None and getattr(self.shard, "step_graph", False) and not self.shard.staged)
        if step_graph and overlap:
            # measured (profiles/r05_s9_step_graph.txt): with the collectives of the two halves issued under two forked streams
            # inside the capture, ProcessGroupNCCL's watchdog thread queries an event recorded in the capturing stream and aborts
            # the process (hipErrorCapturedEvent); the serial replay captures and replays bit-identically
            raise NotImplementedError("HUMANVID_STEP_GRAPH=1 captures the serial replay only: unset HUMANVID_CFG_STREAMS")
        n_steps = len(timesteps) if max_steps is None else min(first_step + max_steps, len(timesteps))
        n_batches = len(list(get_context_scheduler(context_schedule)(0, num_inference_steps, F_, context_frames,
                                                                      context_stride, context_overlap)))
        for i in range(first_step, n_steps):
            t_dev.copy_(t_table[i].expand(rep))
            coeffs.copy_(c_table[i])
            multi = sharded or cfg_split or (self.shard is not None and self.shard.window_groups > 1)
            if use_graph and not multi and i >= first_step + 1:
                if graph is None:
                    # step 0 ran eagerly (allocates every workspace buffer); capture step 1 and replay it
                    graph = self._capture(one_step)
                else:
                    L.call("hv_graph_launch", graph, hvlib.current_stream())
            elif overlap and i >= first_step + 1:
                if recorder is None:
                    recorder = [StepRecorder(L), StepRecorder(L)]
                    fork()
                    for hf in (0, 1):  # record each half on its own stream (executes it as well)
                        with torch.cuda.stream(streams[hf]):
                            self.shard.recorder = recorder[hf]
                            recorder[hf].begin()
                            try:
                                half_step(hf)
                            finally:
                                recorder[hf].end()
                                self.shard.recorder = None
                    join_and_finish()
                else:
                    fork()
                    StepRecorder.replay_interleaved(recorder, streams)
                    join_and_finish()
            elif use_graph and multi and i >= first_step + 1:
                if recorder is None:
                    # frame-sharded: record step 1 as command-list segments cut at every collective
                    recorder = StepRecorder(L)
                    self.shard.recorder = recorder
                    recorder.begin()
                    try:
                        one_step()
                    finally:
                        recorder.end()
                        self.shard.recorder = None
                else:
                    if step_graph and recorder.graph is None:
                        recorder.graph = StepRecorder.capture_step_graph(L, recorder.replay)
                    if recorder.graph is not None:
                        recorder.graph.replay()
                    else:
                        recorder.replay()
            else:
                one_step()
            if step_hook is not None:
                step_hook(i)
            if callback is not None and (n_batches - 1) % callback_steps == 0:
                callback(n_batches - 1, sched.timesteps[i], latents)
        if after_loop is not None:
            after_loop(one_step)  # e.g. bench.py: one more, eagerly launched step inside a launch profile
        if graph is not None:
            torch.cuda.current_stream().synchronize()
            L.call("hv_graph_destroy", graph)
        if recorder is not None:
            torch.cuda.synchronize()
            for r in (recorder if isinstance(recorder, list) else [recorder]):
                r.destroy()
        return latents

    def interpolate_latents(self, latents: torch.Tensor, interpolation_factor: int, device):
        """pipeline_pose2vid_long.py:294-337: k-1 blended latents between consecutive denoised frames (linear or slerp, chosen
        process-wide by src.pipelines.utils.set_tensor_interpolation_method)."""
        from .latent_interp import interpolate_latents

        return interpolate_latents(latents, interpolation_factor, device)

    def _capture(self, fn):
        """Capture `fn`'s launches on a side stream into a HIP graph, launch it once, return the exec."""
        import ctypes

        L = hvlib.load()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            L.call("hv_graph_begin", side.cuda_stream)
            fn()
            gx = ctypes.c_void_p()
            L.call("hv_graph_end", side.cuda_stream, ctypes.byref(gx))
        torch.cuda.current_stream().wait_stream(side)
        L.call("hv_graph_launch", gx, hvlib.current_stream())
        return gx

    # ---- shared front-end of the three __call__ variants -----------------------------------------------
    def _clip_and_reference_pass(self, ref_image, width, height, do_cfg, device):
        """CLIP image embedding, VAE-encoded reference latent, ReferenceNet write pass at t = 0 and bank hand-over
        (pipeline_pose2vid_long.py:380-407, 438-446, 470-480).  Returns (clip_embeds [1,768], reader, writer)."""
        clip_image = _clip_preprocess(ref_image)
        clip_embeds = self.image_encoder(clip_image.to(device, dtype=self.image_encoder.dtype)).image_embeds
        ehs = clip_embeds.unsqueeze(1)
        if do_cfg:
            ehs = torch.cat([torch.zeros_like(ehs), ehs], dim=0)
        writer = ReferenceAttentionControl(self.reference_unet, do_classifier_free_guidance=do_cfg, mode="write",
                                           batch_size=1, fusion_blocks="full")
        reader = ReferenceAttentionControl(self.denoising_unet, do_classifier_free_guidance=do_cfg, mode="read",
                                           batch_size=1, fusion_blocks="full")
        ref_tensor = _pil_to_tensor(ref_image, height, width, normalize=True).to(device=device, dtype=self.vae.dtype)
        ref_latents = self.vae.encode(ref_tensor).latent_dist.mean * 0.18215
        # The reference runs the ReferenceNet on [zero-CLIP, CLIP] (ref_image_latents.repeat(2, ...), :472-474), but in read
        # mode bank entry 0 only ever feeds the attention result that mutual_self_attention.py:178-186 overwrites with plain
        # self-attention: the write pass needs the conditional entry only -> batch 1, one-entry banks.
        self.reference_unet(ref_latents, torch.zeros((), device=device), encoder_hidden_states=ehs[-1:], return_dict=False)
        reader.update(writer)
        return clip_embeds, reader, writer

    @staticmethod
    def _require_gpu_and_plain_sampling(eta, num_images_per_prompt):
        if eta != 0.0:
            raise NotImplementedError("eta > 0 is not supported (the reference always samples with eta = 0)")
        if num_images_per_prompt != 1:
            raise NotImplementedError("num_images_per_prompt > 1 is not supported (the reference scripts always pass 1)")
        if not torch.cuda.is_available():
            raise RuntimeError("the pipeline needs a ROCm GPU: the denoising path has no CPU fallback")
        return torch.device("cuda")

    # ---- reference-compatible entry point ----------------------------------------------------------
    @torch.no_grad()
    def __call__(self, ref_image, pose_images, camera_embedding, width, height, video_length, num_inference_steps,
                 guidance_scale, num_images_per_prompt=1, eta: float = 0.0,
                 generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None,
                 output_type: Optional[str] = "tensor", return_dict: bool = True,
                 callback: Optional[Callable[[int, int, torch.FloatTensor], None]] = None,
                 callback_steps: Optional[int] = 1, context_schedule="uniform", context_frames=24, context_stride=1,
                 context_overlap=4, context_batch_size=1, interpolation_factor=1, **kwargs):
        device = self._require_gpu_and_plain_sampling(eta, num_images_per_prompt)
        do_cfg = guidance_scale > 1.0
        clip_embeds, reader, writer = self._clip_and_reference_pass(ref_image, width, height, do_cfg, device)
        latents = self.prepare_latents(1, self.denoising_unet.in_channels, width, height, video_length, clip_embeds.dtype,
                                       device, generator)
        pose_cond = _pil_to_tensor(list(pose_images), height, width, normalize=False)  # [F,3,H,W]
        pose_cond = pose_cond.permute(1, 0, 2, 3)[None]  # [1,3,F,H,W]
        if not isinstance(camera_embedding, (tuple, list)):  # (K, c2w) camera parameters: on-device Pluecker front-end
            camera_embedding = camera_embedding.to(device=device, dtype=F32)
            assert camera_embedding.ndim == 5
        latents = self.denoise(latents, pose_cond, camera_embedding, clip_embeds, num_inference_steps, guidance_scale,
                               context_schedule=context_schedule, context_frames=context_frames,
                               context_stride=context_stride, context_overlap=context_overlap, callback=callback,
                               callback_steps=callback_steps or 1)
        reader.clear()
        writer.clear()
        if interpolation_factor > 0:  # pipeline_pose2vid_long.py:576-577 (factor 1: unchanged)
            latents = self.interpolate_latents(latents, interpolation_factor, device)
        images = self.decode_latents(latents)
        if output_type == "tensor":
            images = torch.from_numpy(images)
        if not return_dict:
            return images
        return Pose2VideoPipelineOutput(videos=images)


# ------------------------------------------------------------------------------------------------------------------
@dataclass
class Pose2ImagePipelineOutput:
    images: Union[torch.Tensor, np.ndarray]


class Pose2ImagePipeline(Pose2VideoPipeline):
    """Drop-in for /root/reference/src/pipelines/pipeline_pose2img.py:31-376 (BASELINE.json configs[0]: one frame, no
    motion module, e.g. 256x256 with 4 DDIM steps): the same native denoising loop with a single-frame window."""

    def prepare_latents(self, batch_size, num_channels_latents, width, height, dtype, device, generator, latents=None):
        lat = super().prepare_latents(batch_size, num_channels_latents, width, height, 1, dtype, device, generator,
                                      None if latents is None else latents.unsqueeze(2))
        return lat[:, :, 0]

    @torch.no_grad()
    def __call__(self, ref_image, pose_image, camera_embedding, width, height, num_inference_steps, guidance_scale,
                 num_images_per_prompt=1, eta: float = 0.0,
                 generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None,
                 output_type: Optional[str] = "tensor", return_dict: bool = True,
                 callback: Optional[Callable[[int, int, torch.FloatTensor], None]] = None,
                 callback_steps: Optional[int] = 1, **kwargs):
        device = self._require_gpu_and_plain_sampling(eta, num_images_per_prompt)
        do_cfg = guidance_scale > 1.0
        clip_embeds, reader, writer = self._clip_and_reference_pass(ref_image, width, height, do_cfg, device)
        latents = self.prepare_latents(1, self.denoising_unet.in_channels, width, height, clip_embeds.dtype, device,
                                       generator).unsqueeze(2)  # (bs, c, 1, h', w')  pipeline_pose2img.py:267
        pose_cond = _pil_to_tensor(pose_image, height, width, normalize=False).unsqueeze(2)  # [1,3,1,H,W]
        cam = None
        if camera_embedding is not None:
            cam = camera_embedding.to(device=device, dtype=F32).unsqueeze(2)  # [1,6,1,H,W]  (:298)
            assert cam.ndim == 5
        latents = self.denoise(latents, pose_cond, cam, clip_embeds, num_inference_steps, guidance_scale,
                               context_frames=1, context_overlap=0, callback=callback, callback_steps=callback_steps or 1)
        reader.clear()
        writer.clear()
        image = self.decode_latents(latents)  # (b, c, 1, h, w)
        if output_type == "tensor":
            image = torch.from_numpy(image)
        if not return_dict:
            return image
        return Pose2ImagePipelineOutput(images=image)


class Pose2VideoShortPipeline(Pose2VideoPipeline):
    """Drop-in for /root/reference/src/pipelines/pipeline_pose2vid.py:33-458 (class name there: Pose2VideoPipeline): all
    `video_length` frames in ONE UNet forward per step (no context windows), no camera encoder.  The temporal positional
    encoding caps video_length at temporal_position_encoding_max_len, as in the reference."""

    def __init__(self, vae, image_encoder, reference_unet, denoising_unet, pose_guider, scheduler, image_proj_model=None,
                 tokenizer=None, text_encoder=None):
        super().__init__(vae, image_encoder, reference_unet, denoising_unet, pose_guider, None, scheduler,
                         image_proj_model, tokenizer, text_encoder)

    @torch.no_grad()
    def __call__(self, ref_image, pose_images, width, height, video_length, num_inference_steps, guidance_scale,
                 num_images_per_prompt=1, eta: float = 0.0,
                 generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None,
                 output_type: Optional[str] = "tensor", return_dict: bool = True,
                 callback: Optional[Callable[[int, int, torch.FloatTensor], None]] = None,
                 callback_steps: Optional[int] = 1, **kwargs):
        device = self._require_gpu_and_plain_sampling(eta, num_images_per_prompt)
        do_cfg = guidance_scale > 1.0
        clip_embeds, reader, writer = self._clip_and_reference_pass(ref_image, width, height, do_cfg, device)
        latents = self.prepare_latents(1, self.denoising_unet.in_channels, width, height, video_length, clip_embeds.dtype,
                                       device, generator)
        pose_cond = _pil_to_tensor(list(pose_images), height, width, normalize=False).permute(1, 0, 2, 3)[None]
        latents = self.denoise(latents, pose_cond, None, clip_embeds, num_inference_steps, guidance_scale,
                               context_frames=video_length, context_overlap=0, callback=callback,
                               callback_steps=callback_steps or 1)
        reader.clear()
        writer.clear()
        images = self.decode_latents(latents)
        if output_type == "tensor":
            images = torch.from_numpy(images)
        if not return_dict:
            return images
        return Pose2VideoPipelineOutput(videos=images)
