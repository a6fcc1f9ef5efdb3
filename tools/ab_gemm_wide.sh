#!/bin/bash
# Same-box A/B of the opt-in 256 x 320 x 64 wide-tile GEMM kernel (hv_set_tuning(3, 4)) against the default selection
mkdir -p gpurun_out
{
HUMANVID_TUNING=3=4 timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or parts" 2>&1 | tail -1
for m in 1 4; do HV_GEMM_GLDS=$m timeout 300 python tools/microbench.py --only gemm 2>&1 | grep "^gemm" | grep "N=320\|N=640\|LN fold" | awk -v v=$m '{printf "glds=%s %s\n", v, $0}'; done
for rep in 1 2; do for m in 1 4; do HUMANVID_TUNING=3=$m timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('step glds=$m', d['value'], d['ms_per_step'])"; done; done
HUMANVID_TUNING=3=4 timeout 300 python -m pytest tests/test_gpu_fullsize_parity.py -x -q -k config3 2>&1 | tail -1
} | tee gpurun_out/r03_gemm_wide_ab.txt
