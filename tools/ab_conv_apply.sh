#!/bin/bash
# Same-box A/B: GroupNorm apply + SiLU as the convolution's operand prologue (0) vs as its own pass (1)
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "affine" 2>&1 | tail -1
for rep in 1 2; do for m in 0 1; do HUMANVID_CONV_APPLY=$m timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('step conv_apply=$m', d['value'], d['ms_per_step'])"; done; done
timeout 900 python -m pytest tests/test_gpu_fullsize_parity.py -x -q -s -k "config3" 2>&1 | grep -i "nrmse\|passed\|failed" | tail -5
} | tee gpurun_out/r03_conv_apply_ab.txt
