#!/bin/bash
# round 6, session 16: hv_gemm_xs_kernel (X-stationary, K = 320): hardware check + A/B against the 8-wave kernel (HV_TUNE 11=1 / default)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== kernel tests (gemm)"
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "four_wave" 2>&1 | tail -3
for rep in 1 2; do
for t in "11=1" ""; do
echo "== microbench gemm: HV_TUNE=$t"
HV_TUNE="$t" HV_MB_ONLY_L0=1 timeout 600 python tools/microbench.py --only gemm 2>&1 | grep "^gemm"
done
done
} > gpurun_out/r06_s16.txt 2>&1
cat gpurun_out/r06_s16.txt
