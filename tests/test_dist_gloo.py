"""World-size-2 check of the frame-sharded temporal path on CPU (gloo + host-emulated kernels).

The multi-GPU design shards the images of a window along the frame axis; every op is frame-local
except the temporal self-attention of the motion modules, whose K/V are all-gathered
(RCCL over xGMI on the GPU box).  Here two gloo ranks each run
`Runner.temporal_attention_block(..., sharded=True)` on half of the frames and the concatenated
result must equal the unsharded computation (same LayerNorm / positional-encoding / projection
folding, same kernels)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests", "emu"))
sys.path.insert(0, REPO)

B, F, N, C = 2, 4, 4, 320


def _weights(device="cpu"):
    from humanvid_amd import packing

    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(3 * C, C, generator=g) * C**-0.5
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    pe = torch.randn(8, C, generator=g) * 0.3
    wf, cs, bf = packing.fold_layernorm(qkv, None, gamma, beta)
    wo, bo = torch.randn(C, C, generator=g) * C**-0.5, torch.randn(C, generator=g) * 0.1
    w = {"ab.qkv.w": wf, "ab.qkv.colsum": cs, "ab.qkv.bias": bf, "ab.qkv.pe": packing.pe_table(pe, qkv),
         "ab.to_out.0.w": packing.pack_linear(wo), "ab.to_out.0.bias": bo}
    hid = (torch.randn(B, F, N, C, generator=g) + 0.2).to(torch.bfloat16)
    return w, hid


def _runner(w, shard=None):
    import build_emu

    from humanvid_amd import _abi as A
    from humanvid_amd import lib as hvlib
    from humanvid_amd.runner import Runner

    emu = A.HvLibrary(build_emu.build())
    hvlib._LIB = emu  # test-only injection; the product loader never sees the emulator
    hvlib.current_stream = lambda: None
    return Runner(torch.device("cpu"), w, shard=shard)


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from humanvid_amd.runner import FrameShard

    w, hid = _weights()
    shard = FrameShard()
    f0, fl = shard.frame_range(F)
    run = _runner(w, shard)
    local = hid[:, f0:f0 + fl].reshape(B * fl * N, C).contiguous().clone()
    run.temporal_attention_block("ab", local, B, fl, N, sharded=True)
    outs = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(outs, local)
    if rank == 0:
        full = torch.cat([o.view(B, fl, N, C) for o in outs], dim=1)
        torch.save(full, out_path)
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["alltoall", "allgather"])
def test_frame_sharded_temporal_block_matches_unsharded(tmp_path, exchange, monkeypatch):
    """both exchanges of SURVEY.md section 8(e): frames <-> pixels all-to-all around the attention (default), and the
    all-gather of K/V of all frames"""
    monkeypatch.setenv("HUMANVID_TEMPORAL_EXCHANGE", exchange)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out_path = str(tmp_path / "sharded.pt")
    import build_emu

    build_emu.build()
    mp.spawn(_worker, args=(2, port, out_path), nprocs=2, join=True)
    sharded = torch.load(out_path)
    w, hid = _weights()
    run = _runner(w)
    ref = hid.reshape(B * F * N, C).contiguous().clone()
    run.temporal_attention_block("ab", ref, B, F, N, sharded=False)
    ref = ref.view(B, F, N, C)
    err = float((sharded.float() - ref.float()).norm() / ref.float().norm())
    assert err < 4e-3, err
    assert not torch.equal(ref, hid)  # the block did something


def _group_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from humanvid_amd.runner import FrameShard

    shard = FrameShard(window_groups=2)  # 4 ranks -> 2 window groups of 2 frame-sharding ranks
    assert (shard.world, shard.rank, shard.window_group) == (2, rank % 2, rank // 2)
    w, hid = _weights()
    hid = (hid.float() * (1.0 + 0.5 * shard.window_group)).to(torch.bfloat16)  # each group works on its own window
    f0, fl = shard.frame_range(F)
    run = _runner(w, shard)
    local = hid[:, f0:f0 + fl].reshape(B * fl * N, C).contiguous().clone()
    run.temporal_attention_block("ab", local, B, fl, N, sharded=True)   # all-to-all inside the sub-group only
    torch.save(local.view(B, fl, N, C), os.path.join(out_dir, f"r{rank}.pt"))
    acc = torch.full((3,), float(rank + 1))
    shard.all_reduce(acc)                                                # the accumulator: over ALL ranks of the job
    assert torch.equal(acc, torch.full((3,), 10.0)), acc
    dist.destroy_process_group()


def test_window_groups_shard_inside_and_reduce_across(tmp_path):
    """window-parallel x frame-shard on 4 ranks (last bullet): two groups of two ranks, each
    group exchanges its own window's q|k|v inside the group, the noise accumulator is reduced over all four"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import build_emu

    build_emu.build()
    mp.spawn(_group_worker, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    w, hid0 = _weights()
    run = _runner(w)
    for grp in range(2):
        got = torch.cat([torch.load(os.path.join(str(tmp_path), f"r{2 * grp + r}.pt")) for r in range(2)], dim=1)
        hid = (hid0.float() * (1.0 + 0.5 * grp)).to(torch.bfloat16)
        ref = hid.reshape(B * F * N, C).contiguous().clone()
        run.temporal_attention_block("ab", ref, B, F, N, sharded=False)
        ref = ref.view(B, F, N, C)
        err = float((got.float() - ref.float()).norm() / ref.float().norm())
        assert err < 4e-3, (grp, err)


def _cfg_group_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from humanvid_amd import ops
    from humanvid_amd.runner import FrameShard

    shard = FrameShard(cfg_groups=2)  # 4 ranks -> CFG half h = rank // 2, two frame-sharding ranks per half
    assert (shard.world, shard.rank, shard.cfg_group, shard.all_world) == (2, rank % 2, rank // 2, 4)
    with pytest.raises(NotImplementedError):
        FrameShard(cfg_groups=2, window_groups=2)
    # one guided step's accumulator plumbing on the emulated kernels: every rank adds ITS half's prediction for ITS frames,
    # the window counter is added once per frame (by half 0's ranks), one all-reduce over all four ranks, then hv_cfg_ddim_step
    run = _runner(_weights()[0], shard)  # (injects the emulated library)
    L = run.lib
    Cc, Fr, hh, ww = 4, 4, 4, 4
    g = torch.Generator().manual_seed(7)
    lat = torch.randn(1, Cc, Fr, hh, ww, generator=g)
    pred = torch.randn(2, Fr, hh, ww, 8, generator=g).to(torch.bfloat16)  # [half][frame] rows, 4 of 8 padded channels used
    f0, fl = shard.frame_range(Fr)
    frames = torch.arange(f0, f0 + fl, dtype=torch.int32)
    acc_cnt = torch.zeros(2 * Cc * Fr * hh * ww + Fr)
    acc, counter = acc_cnt[:2 * Cc * Fr * hh * ww].view(2, Cc, Fr, hh, ww), acc_cnt[2 * Cc * Fr * hh * ww:]
    spare = torch.zeros_like(counter)
    hf = shard.cfg_group
    ops.accumulate_window(L, None, pred[hf, f0:f0 + fl].contiguous(), 1, Cc, frames, acc[hf:hf + 1], counter if hf == 0 else spare)
    shard.all_reduce(acc_cnt)
    coeffs = torch.tensor([3.5, 0.6, 0.8, 0.7, 0.71414], dtype=torch.float32)
    out = lat.clone()
    ops.cfg_ddim_step(L, None, out, acc, counter, 2, coeffs)
    torch.save(out, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_cfg_groups_split_the_guided_step(tmp_path):
    """CFG-parallel x frame-shard on 4 ranks (round 6): two sub-groups, one per CFG half, each sharding its frames over two
    ranks; the accumulator all-reduce over all four delivers both halves everywhere -- every rank's latents after the fused
    CFG + DDIM step equal the single-process step bit for bit"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import build_emu

    build_emu.build()
    mp.spawn(_cfg_group_worker, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    from humanvid_amd import ops

    run = _runner(_weights()[0])
    L = run.lib
    Cc, Fr, hh, ww = 4, 4, 4, 4
    g = torch.Generator().manual_seed(7)
    lat = torch.randn(1, Cc, Fr, hh, ww, generator=g)
    pred = torch.randn(2, Fr, hh, ww, 8, generator=g).to(torch.bfloat16)
    acc, counter = torch.zeros(2, Cc, Fr, hh, ww), torch.zeros(Fr)
    ops.accumulate_window(L, None, pred.reshape(2 * Fr, hh, ww, 8), 2, Cc, torch.arange(Fr, dtype=torch.int32), acc, counter)
    ref = lat.clone()
    ops.cfg_ddim_step(L, None, ref, acc, counter, 2, torch.tensor([3.5, 0.6, 0.8, 0.7, 0.71414], dtype=torch.float32))
    for r in range(4):
        got = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))
        assert torch.equal(got, ref), (r, float((got - ref).abs().max()))


class _ToyVae(torch.nn.Module):
    """stand-in for AutoencoderKL.decode: frames are independent batch items (GroupNorm is per sample)"""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.conv = torch.nn.Conv2d(4, 3, 3, padding=1)
        self.norm = torch.nn.GroupNorm(1, 3)
        with torch.no_grad():
            self.conv.weight.copy_(torch.randn(3, 4, 3, 3, generator=g) * 0.2)
            self.conv.bias.zero_()
        self.dtype = torch.float32

    def decode(self, z):
        class _Out:
            pass

        o = _Out()
        o.sample = self.norm(torch.nn.functional.interpolate(self.conv(z), scale_factor=2.0))
        return o


def _decode_worker(rank, world, port, out_path, window_groups=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from humanvid_amd.pipeline import Pose2VideoPipeline

    pipe = Pose2VideoPipeline(_ToyVae(), None, None, None, None, None, None).enable_frame_sharding(window_groups=window_groups)
    assert pipe.shard.all_world == world and pipe.shard.world == world // window_groups
    lat = torch.randn(1, 4, 6, 8, 8, generator=torch.Generator().manual_seed(9))
    video = pipe.decode_latents(lat, frames_per_batch=2)
    if rank == 1:
        torch.save(torch.from_numpy(video), out_path)
    dist.destroy_process_group()


def test_vae_decode_batched_and_frame_sharded_matches_per_frame_loop(tmp_path):
    """decode scheduling (batched calls, frames split over ranks) must reproduce the
    reference's serial per-frame loop (pipeline_pose2vid_long.py:114-127) exactly."""
    from humanvid_amd.pipeline import Pose2VideoPipeline

    vae = _ToyVae()
    lat = torch.randn(1, 4, 6, 8, 8, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        ref = torch.cat([vae.decode(1 / 0.18215 * lat[:, :, i]).sample for i in range(6)])
    ref = (ref.view(1, 6, 3, 16, 16).permute(0, 2, 1, 3, 4) / 2 + 0.5).clamp(0, 1)
    pipe = Pose2VideoPipeline(vae, None, None, None, None, None, None)
    for fpb in (1, 4, 8):
        got = torch.from_numpy(pipe.decode_latents(lat, frames_per_batch=fpb))
        assert torch.allclose(got, ref, atol=1e-6), fpb
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out_path = str(tmp_path / "video.pt")
    mp.spawn(_decode_worker, args=(2, port, out_path), nprocs=2, join=True)
    assert torch.allclose(torch.load(out_path), ref, atol=1e-6)
    # window-parallel mode (two sub-groups of one rank): the decode still splits over BOTH ranks
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out_path = str(tmp_path / "video_groups.pt")
    mp.spawn(_decode_worker, args=(2, port, out_path, 2), nprocs=2, join=True)
    assert torch.allclose(torch.load(out_path), ref, atol=1e-6)
