#!/bin/bash
# Same-box A/B of the GEMM k-loop schedule: groups issued half a step .. one step ahead (base) vs one and a half steps ahead
mkdir -p gpurun_out
V="tools/bin/lib_gemm_base.so humanvid_amd/lib/libhumanvid_hip.so"
{
timeout 120 tools/bin/hwcheck 2>&1 | tail -12
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or parts" 2>&1 | tail -2
for v in $V; do HV_LIB=$v timeout 300 python tools/microbench.py --only gemm 2>&1 | grep "^gemm" | awk -v v=$(basename $v) '{printf "%-22s %s\n", v, $0}'; done
for rep in 1 2; do for v in $V; do HUMANVID_HIP_LIB=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('step $v', d['value'], d['ms_per_step'])"; done; done
} | tee gpurun_out/r03_gemm_deep_ab.txt
