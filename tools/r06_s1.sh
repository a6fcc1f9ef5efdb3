#!/bin/bash
# round 6, session 1: hv_gemm_w4_kernel (hv_gemm4.h, first version: new k-loop, the existing epilogues) -- hardware check
# (GEMM kernel tests) and a same-box A/B against the 8-wave 256 x 256 kernel (HV_TUNE 10=0).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== kernel tests (gemm)"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" 2>&1 | tail -5
for rep in 1 2; do
echo "== microbench gemm: four-wave kernel (default)"
timeout 600 python tools/microbench.py --only gemm 2>&1 | grep "^gemm"
echo "== microbench gemm: 8-wave kernel (HV_TUNE 10=0)"
HV_TUNE="10=0" timeout 600 python tools/microbench.py --only gemm 2>&1 | grep "^gemm"
done
} > gpurun_out/r06_s1.txt 2>&1
tail -70 gpurun_out/r06_s1.txt
