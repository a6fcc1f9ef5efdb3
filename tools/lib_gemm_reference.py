"""What the vendor library reaches on the GEMM shapes of a denoising step (measurement aid, not a product path: the
product's GEMMs are the hand-written kernels of humanvid_amd/csrc/hv_gemm.h; this prints torch.matmul = hipBLASLt / rocBLAS
on the same M x N x K in bf16, without any of the fused epilogues, as a yardstick for the k-loop).

    python tools/lib_gemm_reference.py
"""
import torch

SHAPES = [  # (label, M, N, K) -- the shapes of profiles/r04_*_step_profile.tsv at 24f x 768x512, guidance on
    ("ff1 level 0 (GEGLU)", 294912, 2560, 320),
    ("ff1 level 1 (GEGLU)", 73728, 5120, 640),
    ("ff1 level 2 (GEGLU)", 18432, 10240, 1280),
    ("ff2 level 0", 294912, 320, 1280),
    ("ff2 level 1", 73728, 640, 2560),
    ("ff2 level 2", 18432, 1280, 5120),
    ("qkv level 0", 294912, 960, 320),
    ("qkv level 1", 73728, 1920, 640),
    ("qkv level 2", 18432, 3840, 1280),
    ("proj level 0", 294912, 320, 320),
    ("proj level 1", 73728, 640, 640),
    ("proj level 2", 18432, 1280, 1280),
]


def main():
    dev = torch.device("cuda:0")
    for label, M, N, K in SHAPES:
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            torch.matmul(x, w.t(), out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            torch.matmul(x, w.t(), out=y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("library %-22s M=%-6d N=%-5d K=%-4d %8.3f ms %8.1f TF/s %8.1f GB/s" % (
            label, M, N, K, ms, 2.0 * M * N * K / ms * 1e-9, 2.0 * (M * K + N * K + M * N) / ms * 1e-6), flush=True)


if __name__ == "__main__":
    main()
