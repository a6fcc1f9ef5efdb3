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


def test_frame_sharded_temporal_block_matches_unsharded(tmp_path):
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
