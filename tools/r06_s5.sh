#!/bin/bash
# round 6, session 5: deferred stores spread over five k-tiles + one-round-trip epilogue loads: tests, A/B, trace
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== kernel tests (gemm)"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" 2>&1 | tail -3
for rep in 1 2; do
for t in "" "10=3" "10=0"; do
echo "== microbench gemm: HV_TUNE=$t"
HV_TUNE="$t" timeout 600 python tools/microbench.py --only gemm 2>&1 | grep "^gemm" | grep "qkv\|ff1"
done
done
echo "== trace build"
HV_LIB=tools/bin/lib_w4trace.so timeout 600 python tools/microbench.py --only gemm 2>&1 | grep -A1 "^gemm qkv\|^gemm ff1"
} > gpurun_out/r06_s5.txt 2>&1
cat gpurun_out/r06_s5.txt
