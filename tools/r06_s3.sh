#!/bin/bash
# round 6, session 3: hv_gemm_w4_kernel with deferred output stores (192 x 256 x 64 tiles) -- hardware check and same-box A/B:
# default (deferred) / HV_TUNE 10=2 (four waves, stores at once) / 10=0 (8-wave kernel), interleaved, two rounds.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== kernel tests (gemm)"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" 2>&1 | tail -5
for rep in 1 2; do
for t in "" "10=2" "10=0"; do
echo "== microbench gemm: HV_TUNE=$t"
HV_TUNE="$t" timeout 600 python tools/microbench.py --only gemm 2>&1 | grep "^gemm" | grep "qkv\|ff1"
done
done
} > gpurun_out/r06_s3.txt 2>&1
cat gpurun_out/r06_s3.txt
