#!/bin/bash
# round 6, session 4: where a workgroup's time goes in hv_gemm_w4_kernel (timing build: tools/build_variant.sh w4trace k_gemm -DHV_W4_TRACE)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for t in "" "10=2"; do
echo "== HV_TUNE=$t"
HV_LIB=tools/bin/lib_w4trace.so HV_TUNE="$t" timeout 600 python tools/microbench.py --only gemm 2>&1 | grep -A1 "^gemm qkv\|^gemm ff1"
done
} > gpurun_out/r06_s4_w4_trace.txt 2>&1
cat gpurun_out/r06_s4_w4_trace.txt
