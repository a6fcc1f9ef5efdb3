#!/bin/bash
# Round 4, GPU call 7: half of a GEGLU tile's output stores parked in LDS and drained under the next tile's MFMAs (256 x 256 x 64).
mkdir -p gpurun_out
{
echo "== kernel tests"; timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm or geglu" 2>&1 | tail -3
timeout 300 python tools/microbench.py --only gemm 2>&1 | grep "^gemm" | grep "ff1"
HUMANVID_TUNING=8=1 timeout 300 python tools/microbench.py --only gemm 2>&1 | grep "^gemm" | grep "ff1" | sed 's/^/pp (no deferral) /'
for rep in 1 2 3; do timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('step', round(d['value'],3), round(d['ms_per_step'],2))"; done
tools/bin/gemm_trace 294912 2560 320 2 1 0 | tail -2
} 2>&1 | cut -c1-400 | tee gpurun_out/r04_s7.txt
