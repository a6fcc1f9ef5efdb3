"""Fixed cost of one collective on this software stack, one rank (every collective a local copy): torch.distributed on a
one-rank `nccl` group (collective on the process group's own stream, two event hops to / from the caller's stream) against
RCCL called directly through ctypes on the caller's stream (ncclAllToAll / ncclAllReduce), eagerly and inside a captured
graph.  GPU time per collective from events around a train of N collectives, each followed by a tiny dependent kernel
(the shape of the sharded step: kernel, collective, kernel, ...).

    python tools/rccl_latency.py
"""
import ctypes
import os
import time

import torch
import torch.distributed as dist


class NcclUniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_byte * 128)]


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    rccl = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
    uid = NcclUniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, NcclUniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    rccl.ncclAllToAll.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    NCCL_BF16 = 9  # ncclBfloat16

    N = 100
    for nbytes in (64 * 1024, 8 << 20, 64 << 20):
        n = nbytes // 2
        a = torch.randn(n, device=dev).bfloat16()
        b = torch.empty_like(a)
        st = torch.cuda.current_stream()

        def touch():
            b[:16].add_(1.0)  # a tiny dependent kernel behind the collective

        def via_torch():
            dist.all_to_all_single(b, a)
            touch()

        def via_rccl():
            rc = rccl.ncclAllToAll(a.data_ptr(), b.data_ptr(), n, NCCL_BF16, comm, ctypes.c_void_p(st.cuda_stream))
            assert rc == 0, rc
            touch()

        def copy_only():
            b.copy_(a)
            touch()

        res = {}
        for name, fn in (("device copy", copy_only), ("torch.distributed", via_torch), ("RCCL on the caller's stream", via_rccl)):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(N):
                fn()
            e1.record()
            host = (time.perf_counter() - t0) / N * 1e6
            torch.cuda.synchronize()
            res[name] = (e0.elapsed_time(e1) / N * 1e3, host)
        # the same train captured into one graph (RCCL supports stream capture)
        try:
            g = torch.cuda.CUDAGraph()
            s2 = torch.cuda.Stream()
            with torch.cuda.stream(s2):
                def via_rccl_s2():
                    rc = rccl.ncclAllToAll(a.data_ptr(), b.data_ptr(), n, NCCL_BF16, comm, ctypes.c_void_p(s2.cuda_stream))
                    assert rc == 0, rc
                    touch()
                via_rccl_s2()
                torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=s2):
                    for _ in range(N):
                        via_rccl_s2()
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            res["RCCL inside a captured graph"] = (e0.elapsed_time(e1) / N * 1e3, 0.0)
        except Exception as e:  # noqa: BLE001
            res["RCCL inside a captured graph"] = (float("nan"), 0.0)
            print("graph capture of the RCCL call failed:", repr(e)[:200])
        for name, (gpu_us, host_us) in res.items():
            print("%9d bytes  %-32s %8.1f us per collective + kernel on the GPU timeline, %6.1f us of host time" % (nbytes, name, gpu_us, host_us), flush=True)
    os._exit(0)  # (tearing down a communicator that took part in a graph capture hangs on this stack: leave without it)


if __name__ == "__main__":
    main()
