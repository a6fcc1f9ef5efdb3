#!/bin/bash
# round 6, session 8: one-round-trip per-row loads in the 128 x 128 kernel's permuted epilogue (build variant -DHV_EPI_G4=1): microbench + step
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for rep in 1 2; do
for lib in "" tools/bin/lib_g4.so; do
echo "== lib=$lib"
HV_LIB=$lib timeout 600 python tools/microbench.py --only gemm 2>&1 | grep "^gemm" | grep "proj\|ff2\|residual"
done
done
for rep in 1 2; do
for lib in "" tools/bin/lib_g4.so; do
echo "== step lib=$lib"
HUMANVID_HIP_LIB=$lib timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
done
} > gpurun_out/r06_s8.txt 2>&1
cat gpurun_out/r06_s8.txt
