"""L2-miss read bytes of single GEMM shapes (FETCH_SIZE), in two steps:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_gf -- python tools/gemm_fetch.py run
    python tools/gemm_fetch.py parse /tmp/pmc_gf
`run` launches each shape of SHAPES exactly REPS times in order (nothing else on the LDS-DMA GEMM kernels); `parse` averages the
dispatches of the GEMM kernels in order, REPS at a time (KiB x 1024 x 2: the corrections of tools/pmc_by_shape.py)."""
import csv
import glob
import os
import sys

SHAPES = [  # (label, M, N, K, geglu)
    ("ff1 level 0", 294912, 2560, 320, 1),
    ("ff1 level 1", 73728, 5120, 640, 1),
    ("ff1 level 2", 18432, 10240, 1280, 1),
    ("qkv level 0", 294912, 960, 320, 0),
    ("qkv level 1", 73728, 1920, 640, 0),
    ("qkv level 2", 18432, 3840, 1280, 0),
]
REPS = 3


def run():
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from humanvid_amd import _abi as A, lib as hvlib, ops
    L = A.HvLibrary(os.environ["HV_LIB"]) if os.environ.get("HV_LIB") else hvlib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for label, M, N, K, geglu in SHAPES:
        x = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        b = torch.zeros(N, device=dev)
        y = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=torch.bfloat16)
        kw = {}
        if geglu:  # the LayerNorm-fold + GEGLU form of the step
            kw = dict(row_mean=torch.zeros(M, device=dev), row_rstd=torch.ones(M, device=dev), colsum=torch.zeros(N, device=dev), geglu=True)
        for _ in range(REPS):
            ops.gemm(L, st, x, w, y, M=M, bias=b, **kw)
        torch.cuda.synchronize()


def parse(d):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    rows = [r for r in csv.DictReader(open(files[0])) if r["Counter_Name"] == "FETCH_SIZE" and "hv_gemm_glds" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for i, (label, M, N, K, geglu) in enumerate(SHAPES):
        part = rows[i * REPS:(i + 1) * REPS]
        if len(part) < REPS:
            break
        mb = sum(float(r["Counter_Value"]) for r in part) / REPS * 2048.0 / 1e6
        alg = 2.0 * (M * K + N * K) / 1e6
        print("%-12s M=%-6d N=%-5d K=%-4d fetched %8.1f MB per launch, algorithmic reads %7.1f MB  (%.2fx)" % (label, M, N, K, mb, alg, mb / alg))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else parse(sys.argv[2])
