#!/usr/bin/env python3
"""Headline benchmark: denoising steps / second, 24f x 768 x 512 Pose2Video (BASELINE.json config #3),
CFG 3.5, SD-1.5 geometry + AnimateDiff motion modules + CameraCtrl Pluecker encoder, bf16 storage /
fp32 accumulation, random-init weights and synthetic latents/pose/camera/CLIP/bank tensors of the
named shapes (no checkpoints or datasets are reachable).

    python bench.py --gpus N --steps K --warmup W [--config 2|3|5]
    (N > 1: launched by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`,
     or directly -- bench.py then re-executes itself under torch.distributed.run with one rank per GPU)

A "step" = one pass of the hot path: pack latents -> UNet3D forward on both CFG halves (ReferenceNet
banks injected, pose+camera conditioning added) -> window accumulation -> CFG + DDIM update.  With
N > 1 the frames of each window are sharded over the ranks (frames<->pixels all-to-all around every
temporal attention over RCCL/xGMI; HUMANVID_TEMPORAL_EXCHANGE=allgather replicates K/V instead): the
total work is fixed, so scaling is "strong".  Rank 0 prints ONE JSON line; its `roofline` block is
built from the real launches of one profiled step (hv_profile_begin/end).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def build_models(dev):
    from humanvid_amd.arch import SD15_INFERENCE_V2
    from humanvid_amd.conditioning import CameraPoseEncoder, PoseGuider
    from humanvid_amd.unet3d import UNet3DConditionModel

    torch.manual_seed(0)
    with torch.device(dev):
        unet = UNet3DConditionModel(**SD15_INFERENCE_V2)
        pg = PoseGuider(320, block_out_channels=(16, 32, 96, 256))
        cam = CameraPoseEncoder(downscale_factor=8, channels=[320], nums_rb=2, cin=384, ksize=1, sk=True, use_conv=False,
                                compression_factor=1, temporal_attention_nhead=8, attention_block_types=["Temporal_Self"],
                                temporal_position_encoding=True, temporal_position_encoding_max_len=24)
    with torch.no_grad():  # zero-initialised branches would make whole sub-graphs trivial: re-randomise
        for mod in (unet, pg, cam):
            for name, p in mod.named_parameters():
                if p.ndim >= 2 and float(p.abs().max()) == 0.0:
                    p.normal_(0.0, 0.02)
    return unet, pg, cam


# ---------------------------------------------------------------------------------------------------------------------
# roofline from the step's REAL launches: one extra denoising step is launched eagerly inside an hv_profile_begin/end
# bracket (HIP events around every kernel on the launch stream); the library files each launch under
# "kernel variant | shape" and this module prices the shapes (algorithmic FLOPs and bytes per DESIGN.md section 3).
def _kv(shape: str):
    out = {}
    for tok in shape.split():
        k, _, v = tok.partition("=")
        try:
            out[k] = int(v)
        except ValueError:  # a launch note this module does not price must not take the bench line down
            out[k] = v
    return out


def price_launch(key: str):
    """-> (flops, algorithmic bytes) of ONE launch described by a profile key"""
    kern, _, shape = key.partition(" | ")
    a = _kv(shape) if shape else {}
    if kern.startswith("hv_gemm"):
        n_out = a["N"] // 2 if a["geglu"] else a["N"]
        fl = 2.0 * a["M"] * a["N"] * a["K"]
        by = 2.0 * (a["M"] * a["K"] + a["N"] * a["K"]) + a["M"] * n_out * (4.0 if a["f32"] else 2.0) \
            + (2.0 * a["M"] * n_out if a["res"] else 0.0)
        return fl, by
    if kern.startswith(("hv_conv3x3", "hv_conv_w4")):
        px_o, px_i = a["n"] * a["Ho"] * a["Wo"], a["n"] * a["Hs"] * a["Ws"]
        fl = 2.0 * 9 * a["Cin"] * a["Cout"] * px_o
        by = 2.0 * (px_i * a["Cin"] + 9 * a["Cin"] * a["Cout"] + px_o * a["Cout"] * (2 if a["res"] else 1))
        return fl, by
    if kern.startswith("hv_attention_fp8_amax") or kern.startswith("hv_attention_fp8_quant"):
        C = a["heads"] * int(kern.split("<")[1].split(">")[0])
        return 0.0, (4.0 if "amax" in kern else 6.0) * a["n"] * a["L"] * C  # K and V^T read (+ e4m3 copies written)
    if kern.startswith("hv_attention"):
        C = a["heads"] * a["D"]
        banked = a["n"] / 2 if a["bank"] else 0  # CFG layout: the unconditional half attends its own keys only
        fl = 4.0 * C * a["Lq"] * (a["L1"] * a["n"] + a["L2"] * banked)
        by = 2.0 * C * (2 * a["n"] * a["Lq"] + 2 * a["n"] * a["L1"] + (4 * a["L2"] if a["bank"] else 0))
        return fl, by
    if kern.startswith("hv_temporal"):
        D = int(kern.split("<")[1].split(">")[0])
        C = 8 * D
        fl = 4.0 * a["Fq"] * a["Fkv"] * C * a["B"] * a["P"]
        by = 2.0 * C * a["B"] * a["P"] * (2 * a["Fq"] + 2 * a["Fkv"])
        return fl, by
    if kern.startswith("hv_gn_partial"):
        return 0.0, 2.0 * a["n"] * a["pixels"] * a["C"]
    if kern.startswith("hv_ln_stats"):
        return 0.0, 2.0 * a["M"] * a["C"]
    if kern.startswith("hv_affine_apply") and "rows" in a:
        return 0.0, 4.0 * a["rows"] * (a["C"] + a.get("C2", 0))  # bf16 in, bf16 out
    return 0.0, 0.0


def price_launch_rw(key: str):
    """-> (algorithmic bytes read, written) of ONE launch: the split of price_launch's byte count that the per-shape counter
    table (tools/pmc_by_shape.py) compares FETCH_SIZE / WRITE_SIZE with"""
    kern, _, shape = key.partition(" | ")
    a = _kv(shape) if shape else {}
    _, total = price_launch(key)
    if kern.startswith("hv_gemm"):
        n_out = a["N"] // 2 if a["geglu"] else a["N"]
        wr = a["M"] * n_out * (4.0 if a["f32"] else 2.0)
    elif kern.startswith(("hv_conv3x3", "hv_conv_w4")):
        wr = 2.0 * a["n"] * a["Ho"] * a["Wo"] * a["Cout"]
    elif kern.startswith("hv_attention_fp8_quant"):
        wr = 2.0 * a["n"] * a["L"] * a["heads"] * int(kern.split("<")[1].split(">")[0])
    elif kern.startswith("hv_attention") and "amax" not in kern:
        wr = 2.0 * a["heads"] * a["D"] * a["n"] * a["Lq"]
    elif kern.startswith("hv_temporal"):
        wr = 2.0 * 8 * int(kern.split("<")[1].split(">")[0]) * a["B"] * a["P"] * a["Fq"]
    elif kern.startswith("hv_affine_apply") and "rows" in a:
        wr = total / 2
    else:
        wr = 0.0
    return total - wr, wr


def csrc_digest() -> str:
    """content hash of the kernel sources: what a committed counter summary is valid for (the GPU box has no .git)"""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(REPO, "humanvid_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def profile_step(run_step):
    """run_step() launches one denoising step eagerly on the current stream -> per-kernel-variant totals"""
    import ctypes

    from humanvid_amd import lib as hvlib

    L = hvlib.load()
    torch.cuda.synchronize()
    L.call("hv_profile_begin")
    try:
        run_step()
    finally:
        buf = ctypes.create_string_buffer(1 << 20)
        need = L.cdll.hv_profile_end(buf, len(buf))
        if need > len(buf):  # the text is kept until a buffer of the returned size collects it
            buf = ctypes.create_string_buffer(need)
            need = L.cdll.hv_profile_end(buf, len(buf))
    if need < 0:
        raise RuntimeError("hv_profile_end failed")
    kernels = {}
    if os.environ.get("HV_PROFILE_DUMP"):  # raw per-(kernel, shape) lines for tools / profiles
        with open(os.environ["HV_PROFILE_DUMP"], "w") as fh:
            fh.write("launches\ttotal_ms\tkernel | shape\n" + buf.value.decode())
    for line in buf.value.decode().splitlines():
        if line.startswith("#"):  # "#seq": launch order, for tools/pmc_by_shape.py
            continue
        cnt, ms, key = line.split("\t", 2)
        fl, by = price_launch(key)
        k = kernels.setdefault(key.partition(" | ")[0], dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
        k["launches"] += int(cnt)
        k["ms"] += float(ms)
        k["flops"] += fl * int(cnt)
        k["bytes"] += by * int(cnt)
    return kernels


def roofline_from_profile(kernels, traffic_file):
    total_ms = sum(k["ms"] for k in kernels.values())
    name, dom = max(kernels.items(), key=lambda kv: kv[1]["ms"])
    tf = dom["flops"] / dom["ms"] / 1e9
    roof = {"bound": "mfma", "achieved": tf, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_BF16_TFLOPS,
            "traffic": None, "kernel": name, "launches_per_step": dom["launches"], "kernel_ms_per_step": dom["ms"],
            "avg_launch_ms": dom["ms"] / dom["launches"], "flops_per_launch": dom["flops"] / dom["launches"],
            "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"],
            "source": "hv_profile_begin/end: HIP events around every launch of one eagerly launched step after the timed "
                      "region (real epilogues, variants and multiplicities); shapes priced by bench.price_launch",
            "sum_of_kernel_ms_per_step": total_ms}
    # roofline.traffic: HBM-side bytes per launch of the dominant kernel from the committed counter passes (PMC counters
    # cannot be read inside the timed run).  A summary made from OTHER kernel sources is refused: the file carries the
    # content hash of humanvid_amd/csrc it was collected with (tools/pmc_by_shape.py).
    if os.path.exists(traffic_file):
        t = json.load(open(traffic_file))
        if t.get("csrc_digest") != csrc_digest():
            roof["traffic_note"] = (f"refused {os.path.basename(traffic_file)}: collected with kernel sources "
                                    f"{t.get('csrc_digest')}, this build is {csrc_digest()} (re-run tools/final_r06.sh)")
        else:
            k = t.get("kernels", {}).get(name)
            if k is not None:
                roof["traffic"] = k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"]
                roof["traffic_detail"] = dict(k, source=t.get("source"), corrections=t.get("corrections"),
                                              csrc_digest=t["csrc_digest"])
                roof["traffic_by_shape"] = [s_ for s_ in t.get("shapes", []) if s_["key"].startswith(name + " | ")]
    table = []
    for n, k in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"])[:10]:
        table.append({"kernel": n, "launches": k["launches"], "ms": round(k["ms"], 3),
                      "tflops": round(k["flops"] / k["ms"] / 1e9, 1) if k["ms"] > 0 else 0.0,
                      "algorithmic_GBps": round(k["bytes"] / k["ms"] / 1e6, 1) if k["ms"] > 0 else 0.0,
                      "frac_of_mfma_peak": round(k["flops"] / k["ms"] / 1e9 / PEAK_BF16_TFLOPS, 4) if k["ms"] > 0 else 0.0})
    return roof, table


def cpu_baseline(budget_hw=(48, 32), frames=24):
    """Time the fp32 oracle (a port of the reference's forward) on THIS host's cores: one CFG UNet forward + DDIM update at
    F=24, full SD-1.5 widths.  Default since round 5: the WHOLE config-#3 step (latent 96x64: no
    scaling, ~5.5 min on the bench host's 128 cores).  --cpu-baseline quarter: latent 48x32 = a quarter of the pixels, scaled
    to config #3 by the as-written FLOP ratio (~5.2: the spatial attention is quadratic in the pixels)."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle_torch as O  # test infrastructure: timed here as the reported CPU baseline only

    from humanvid_amd.arch import DEFAULT_UNET3D_CONFIG, SD15_INFERENCE_V2
    from humanvid_amd.workload import unet3d_flops

    cfg = dict(O.SD15_UNET3D_CFG)
    sd = O.make_unet3d_weights(cfg, seed=0)
    h, w = budget_hw
    g = torch.Generator().manual_seed(42)
    lat = torch.randn(1, 4, frames, h, w, generator=g)
    ehs = torch.cat([torch.zeros(1, 1, 768), torch.randn(1, 1, 768, generator=g)])
    pose = torch.randn(1, 320, frames, h, w, generator=g).repeat(2, 1, 1, 1, 1)
    banks = {p: torch.randn(2, (h >> l) * (w >> l), c, generator=g)
             for p in O.transformer_locations(cfg)
             for c, l in [(sd[p + ".norm.weight"].numel(), {320: 0, 640: 1, 1280: 2}[sd[p + ".norm.weight"].numel()])]}
    sched = O.DDIM()
    sched.set_timesteps(30)
    t0 = time.time()
    with torch.no_grad():
        pred = O.unet3d_forward(sd, cfg, lat.repeat(2, 1, 1, 1, 1), 966, ehs, pose, banks, do_cfg=True)
        u, c = pred.chunk(2)
        sched.step(u + 3.5 * (c - u), 966, lat)
    dt = time.time() - t0
    full = dict(DEFAULT_UNET3D_CONFIG)
    full.update(SD15_INFERENCE_V2)
    ratio = unet3d_flops(full, 2, 24, 96, 64, True)["total"] / unet3d_flops(full, 2, frames, h, w, True)["total"]
    import platform

    return dict(value=1.0 / (dt * ratio), unit="steps/s", cores=torch.get_num_threads(), kind="port",
                host=f"the bench host ({platform.node()}, {os.cpu_count()} logical cores)",
                sample=f"fp32 oracle (port of the reference forward), 1 CFG step at F={frames}, latent {h}x{w} "
                       f"({100.0 * h * w / (96 * 64):.0f} % of the config-3 pixels), SD-1.5 widths: {dt:.1f} s"
                       + (f"; scaled by as-written FLOP ratio {ratio:.2f} to 24f x 768x512" if ratio > 1.001 else " (no scaling)"))


CONFIGS = {
    # BASELINE.json configs[] index -> frames, height, width, description
    2: dict(F=16, H=512, W=512, what="Pose2Video 16f x 512x512, static camera (BASELINE.json configs[1])"),
    3: dict(F=24, H=768, W=512, what="Pose2Video 24f x 768x512, CameraCtrl Pluecker embedding (BASELINE.json configs[2])"),
    5: dict(F=48, H=1024, W=576, what="Pose2Video 48f x 1024x576, 3 context windows of 24 per step (BASELINE.json configs[4]; "
                                      "spatial attention bf16 by default, fp8 e4m3 with --fp8-attention 1: a footprint option)"),
}


def _self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU."""
    import socket
    import subprocess

    if torch.cuda.device_count() < args.gpus:
        raise SystemExit(f"--gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL / tensor sharing across processes needs it here
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS), help="BASELINE.json configs[] number (1-based)")
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--fp8-attention", type=int, default=None, choices=[0, 1],
                    help="spatial attention on the fp8 (e4m3) MFMA (what BASELINE.json configs[4] names); default 0 for every "
                         "config: on gfx950 the non-scaled fp8 MFMA runs at the bf16 rate and the kernel pays an extra scaling "
                         "multiply per score, so it is a precision / footprint option, not the fast path (DESIGN.md section 3)")
    ap.add_argument("--window-groups", type=int, default=1,
                    help="N > 1 GPUs: split the ranks into this many groups that take different context windows of a step "
                         "(window-parallel x frame-shard; clips with several windows per step, e.g. --config 5)")
    ap.add_argument("--cfg-groups", default="auto", choices=["auto", "1", "2"],
                    help="N > 1 GPUs: the CFG-parallel axis (FrameShard cfg_groups): 2 = the ranks split into two sub-groups, one per "
                         "CFG half of the guided step (B = 1 forwards; at N = 2 no temporal exchange at all, one accumulator "
                         "all-reduce per step), each sharding its frames over N / 2 ranks; 1 = every rank runs both halves of its "
                         "frames (N-way frame shard); auto = 2 for even N (DESIGN.md section 5)")
    ap.add_argument("--cfg-half", type=int, default=None, choices=[0, 1],
                    help="diagnostic (N = 1 only): time the step ONE rank of a two-rank CFG-parallel job runs -- the B = 1 forward of "
                         "CFG half 0 (unconditional) or 1 (conditional, bank keys), all 24 frames, plus the accumulator all-reduce on "
                         "a one-rank RCCL group; the line is a diagnostic, not the metric")
    ap.add_argument("--cfg-streams", default="0", choices=["auto", "0", "1"],
                    help="N > 1 GPUs: the two CFG halves of a step on two streams, replayed interleaved, so that one half's temporal "
                         "exchange runs under the other half's kernels (DESIGN.md section 5).  Default 0 = the serial replay (the "
                         "overlapped one has never run with more than one rank on the real transport); 1: on; auto: on after a "
                         "three-step probe run outside the timed region -- if the probe raises, the serial path is timed and the "
                         "line says so (a rank that hangs inside a collective is NOT caught by the probe)")
    ap.add_argument("--step-graph", default="0", choices=["0", "1"],
                    help="N > 1 GPUs (or --single-rank-sharded): replay the recorded step -- command-list segments AND the RCCL "
                         "collectives between them -- as ONE captured device graph per step instead of one launch per segment plus "
                         "one torch.distributed call per collective.  Default 0: verified bit for bit on a one-rank RCCL group only")
    ap.add_argument("--single-rank-sharded", action="store_true",
                    help="diagnostic (N = 1 only): run the SHARDED code path -- exchange layouts, graph-replayed command-list "
                         "segments, CFG halves on two streams, every collective through RCCL -- with a process group of one rank: "
                         "what the sharded schedule costs before any byte crosses xGMI (DESIGN.md section 5)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "quarter", "full"],
                    help="sample of the CPU baseline (fp32 oracle port on this host's cores): full = the WHOLE config-3 step (a "
                         "timing, not an extrapolation; ~3.5 min on 128 cores, run after the timed GPU region), quarter = a quarter "
                         "of the config-3 pixels scaled by the as-written FLOP ratio (~1 min).  auto (default) = full when this "
                         "process has >= 64 CPU threads (the bench host: 128), quarter otherwise -- on a small host the whole "
                         "step would take tens of minutes and the line would not be printed inside a driver's time limit")
    ap.add_argument("--no-profile", action="store_true", help="skip the profiled extra step (roofline block)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    fp8_attn = bool(args.fp8_attention)
    os.environ["HUMANVID_ATTENTION_FP8"] = "1" if fp8_attn else "0"  # read by UNet3DEngine at construction
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist

    sharded = world > 1 or args.single_rank_sharded or args.cfg_half is not None
    if args.cfg_half is not None:
        if world != 1 or args.single_rank_sharded:
            raise SystemExit("--cfg-half is an N = 1 diagnostic")
        os.environ.setdefault("MASTER_PORT", "29534")
        os.environ.update(RANK="0", WORLD_SIZE="1")
    if args.single_rank_sharded:
        if world != 1:
            raise SystemExit("--single-rank-sharded is an N = 1 diagnostic")
        os.environ["HUMANVID_SINGLE_RANK_SHARDED"] = "1"  # read by FrameShard
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.update(RANK="0", WORLD_SIZE="1")
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from humanvid_amd.arch import DEFAULT_UNET3D_CONFIG, SD15_INFERENCE_V2
    from humanvid_amd.pipeline import Pose2VideoPipeline
    from humanvid_amd.scheduler import DDIMScheduler, get_context_scheduler
    from humanvid_amd.workload import unet3d_flops

    # (same-box A/Bs of kernel selections: HUMANVID_TUNING="key=value,..." -- humanvid_amd/lib.py applies it at load time)
    cfgsel = CONFIGS[args.config]
    F, H, W = args.frames or cfgsel["F"], args.height or cfgsel["H"], args.width or cfgsel["W"]
    h, w = H // 8, W // 8
    unet, pg, cam = build_models(dev)
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=False, steps_offset=1,
                          prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")
    pipe = Pose2VideoPipeline(None, None, None, unet, pg, cam, sched)
    cfg_groups = 1
    if sharded:
        if args.cfg_groups == "2" or (args.cfg_groups == "auto" and world % 2 == 0 and args.window_groups == 1):
            cfg_groups = 2
        pipe.enable_frame_sharding(window_groups=args.window_groups, cfg_groups=cfg_groups if world > 1 else 1)
        if args.cfg_half is not None:  # one rank of a two-rank CFG-parallel job, on a one-rank group
            pipe.shard.cfg_groups, pipe.shard.cfg_group = 2, args.cfg_half
            pipe.shard.cfg_half_diagnostic = True  # the absent rank's half of the accumulator: a copy of this one's (pipeline.py)
            cfg_groups = 2

    g = torch.Generator().manual_seed(42)
    latents = torch.randn(1, 4, F, h, w, generator=g)
    pose = torch.rand(1, 3, F, H, W, generator=torch.Generator().manual_seed(1))
    plucker = torch.randn(1, 6, F, H, W, generator=torch.Generator().manual_seed(3))
    clip = torch.randn(1, 768, generator=torch.Generator().manual_seed(2))
    from humanvid_amd.unet3d import transformer_locations

    eng = unet.engine() if not sharded else None
    if eng is None:
        from humanvid_amd.engine import UNet3DEngine

        unet._engine = eng = UNet3DEngine(unet, shard=pipe.shard)
    gb = torch.Generator(device=dev).manual_seed(5)
    banks = {}
    for loc in transformer_locations(unet):
        C = eng.w[loc + ".proj_in.w"].shape[0]
        lvl = {320: 0, 640: 1, 1280: 2}[C] if loc != "mid_block.attentions.0" else 3
        banks[loc] = torch.randn(2, (h >> lvl) * (w >> lvl), C, device=dev, generator=gb).half().float()
    eng.set_reference_banks(banks, do_cfg=True)
    eng._banks_from_modules = lambda: None  # banks were installed explicitly (no ReferenceNet pass in the timed path)

    K, Wm = args.steps, args.warmup
    # two set-up steps in front of the W warm-up steps, whatever W is: step 0 runs eagerly and allocates every workspace
    # buffer, step 1 is captured into the HIP graph (N = 1) / recorded as command-list segments (N > 1); from step 2 on a
    # step is a replay -- the steady state the metric is quoted on
    SETUP = 1 if args.no_graph else 2
    n_inf = max(30, SETUP + K + Wm)
    times = {}
    prof = {}

    def sync_barrier():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
            torch.cuda.synchronize()

    def hook(i):
        if i == SETUP + Wm - 1:
            sync_barrier()
            times["t0"] = time.perf_counter()
        if i == SETUP + Wm + K - 1:
            sync_barrier()
            times["t1"] = time.perf_counter()

    def after_loop(one_step):
        if not sharded and not args.no_profile:
            prof["kernels"] = profile_step(one_step)
        if sharded:
            # one more, eagerly launched step outside the timed region with every collective counted and bracketed by
            # events: collectives per step, bytes this rank sends, and the exchange time the step is exposed to
            # (a diagnostic: whatever goes wrong in it must not take the bench line down -- every rank still reaches the
            #  reductions below, with ok = 0)
            sh = pipe.shard
            sync_barrier()
            sh.reset_stats()
            sh.measure = True
            t0 = time.perf_counter()
            try:
                one_step()
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t0) * 1e3
                prof["exchange"] = dict(collectives_per_step=sh.stats["collectives"], bytes_sent_per_rank=sh.stats["bytes_sent"],
                                        exposed_exchange_ms=sh.exposed_ms(), eager_step_ms=wall, ok=1.0)
            except Exception as e:  # noqa: BLE001
                prof["exchange"] = dict(collectives_per_step=0, bytes_sent_per_rank=0, exposed_exchange_ms=0.0, eager_step_ms=0.0,
                                        ok=0.0, error=repr(e)[:200])
            finally:
                sh.measure = False

    cfg_streams = None
    if sharded:
        sh = pipe.shard
        sh.overlap_cfg = args.cfg_streams != "0" and not args.no_graph
        cfg_streams = "on" if sh.overlap_cfg else "off"
        sh.step_graph = args.step_graph == "1" and not args.no_graph
        if sh.overlap_cfg and args.cfg_streams == "auto":
            # probe outside the timed region: eager step, recorded step, interleaved replay -- the path has only ever run on the
            # host-staged transport of the one-GPU tests; a failure here must not take the bench line down
            ok = torch.ones(1, device=dev)
            try:
                pipe.denoise(latents.clone(), pose, plucker, clip, n_inf, 3.5, max_steps=3)
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                ok.zero_()
                cfg_streams = "off (probe failed: " + repr(e)[:160] + ")"
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok) < 1.0:
                sh.overlap_cfg = False
                if cfg_streams == "on":
                    cfg_streams = "off (probe failed on another rank)"
    if SETUP + Wm == 0:
        sync_barrier()
        times["t0"] = time.perf_counter()
    pipe.denoise(latents, pose, plucker, clip, n_inf, 3.5, use_graph=not args.no_graph, max_steps=SETUP + Wm + K, step_hook=hook,
                 after_loop=after_loop)
    elapsed = times["t1"] - times["t0"]
    if sharded:
        tmax = torch.tensor([elapsed], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)

    exchange = None
    if sharded and "exchange" in prof:
        ex = prof["exchange"]
        t = torch.tensor([ex["exposed_exchange_ms"], ex["eager_step_ms"], -ex["ok"]], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ex = dict(ex, ok=bool(float(t[2]) <= -1.0))  # every rank measured
        exchange = dict(ex, exposed_exchange_ms=float(t[0]), eager_step_ms=float(t[1]),
                        note="one EAGERLY launched step after the timed region, max over ranks; exposed = event distance around "
                             "every collective on its stream: an upper bound -- the eager step runs the CFG halves one after "
                             "the other, the timed steps replay them interleaved (cfg_streams) so that one half's exchange "
                             "runs under the other half's kernels")
    if rank == 0:
        cfg = dict(DEFAULT_UNET3D_CONFIG)
        cfg.update(SD15_INFERENCE_V2)
        windows = list(get_context_scheduler("uniform")(0, n_inf, F, 24, 1, 4))
        fl_total = sum(unet3d_flops(cfg, 2, len(c), h, w, False)["total"] for c in windows)
        ms_step = elapsed / K * 1e3
        exch = pipe.shard.exchange if sharded else None
        par = "single GPU" if not sharded else ("ONE rank on the sharded code path (diagnostic: --single-rank-sharded)" if world == 1 else (
            f"frame-sharded x{world}: frames<->pixels all-to-all around every temporal attention (RCCL over xGMI)"
            if exch == "alltoall" else f"frame-sharded x{world}: RCCL all-gather of temporal K/V"))
        if args.cfg_half is not None:
            par = (f"DIAGNOSTIC (--cfg-half {args.cfg_half}): the step ONE rank of a two-rank CFG-parallel job runs -- B = 1 forward of CFG half "
                   f"{args.cfg_half} on all frames + the accumulator all-reduce on a one-rank RCCL group; not the metric")
        elif sharded and world > 1 and cfg_groups == 2:
            fw = world // 2
            par = (f"CFG-parallel x2 (one sub-group of {fw} rank(s) per CFG half, B = 1 forwards, accumulator all-reduce over all {world}"
                   + (": no temporal exchange)" if fw == 1 else f") x frame-sharded x{fw} inside each half ({exch})"))
        out = {
            "metric": f"denoising steps/sec, {F}f x {H}x{W} Pose2Video", "value": K / elapsed, "unit": "steps/s",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16 (fp8 e4m3 QK^T / PV in the spatial attention)" if fp8_attn else "bf16", "data": "synthetic",
            "config": {"workload": cfgsel["what"] + f"; CFG 3.5, SD-1.5 UNet3D + motion modules, {len(windows)} window(s) "
                                   "per step, DDIM v-pred",
                       "parallelism": par, "hip_graph": not args.no_graph,
                       # BASELINE.json configs[4] names fp8 MFMA attention; it is a footprint option here, not the fast path
                       # (QK^T reduces over d = 40 / 80 / 160, the 2x-rate MX fp8 MFMA over K = 128), so
                       # the line says which arithmetic the spatial attention of THIS run used
                       "attention_dtype": "fp8 e4m3 (hv_attention_fp8)" if fp8_attn else "bf16",
                       **({} if cfg_streams is None else {"cfg_streams": cfg_streams,
                                                           "step_graph": bool(pipe.shard.step_graph)})},
            "step_algorithmic_tflop": fl_total / 1e12,
            "step_tflops_per_gpu": fl_total / 1e12 / (ms_step / 1e3) / world,
            "step_frac_of_mfma_peak": fl_total / 1e12 / (ms_step / 1e3) / world / PEAK_BF16_TFLOPS,
        }
        if exchange is not None:
            out["exchange"] = exchange
        if "kernels" in prof:
            # the newest committed counter summary (profiles/rNN_pmc_traffic.json); one made from other kernel sources is refused
            import glob

            tfiles = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
            roof, table = roofline_from_profile(prof["kernels"], tfiles[-1] if tfiles else "")
            out["roofline"] = roof
            out["kernels"] = table
        if not sharded and not args.no_cpu_baseline:
            full_step = args.cpu_baseline == "full" or (args.cpu_baseline == "auto" and torch.get_num_threads() >= 64)
            out["cpu_baseline"] = cpu_baseline((96, 64) if full_step else (48, 32))
            ref = os.path.join(REPO, "tests", "golden", f"cpu_reference_config{args.config}.json")
            if os.path.exists(ref):  # the reference SOURCE timed in the build container (oracle/gen_fullsize_golden.py)
                r = json.load(open(ref))
                out["cpu_reference"] = {"value": r["steps_per_s"], "unit": "steps/s", "cores": r["cores"], "kind": "reference",
                                        "host": "the build container (NOT the bench host)",
                                        "sample": "one steady-state denoising step of /root/reference's own code (fp32, "
                                                  f"{r['cores']} vCPU build container), committed measurement: "
                                                  f"tests/golden/cpu_reference_config{args.config}.json"}
        print(json.dumps(out))
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
