#!/bin/bash
# Round 4, GPU call 4 (~7 GPU-minutes): the optimistic reference maximum of hv_attention40 (careful second pass only after an
# overflow), the convolution's patch-major raster as the default for the deep levels.
#   gpurun --timeout 900 -- 'bash tools/r04_s4.sh'
mkdir -p gpurun_out
{
echo "== kernel tests"; timeout 500 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention or conv" 2>&1 | tail -3
echo "== full-size parity (config 3 forward, loop body)"; timeout 500 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_fullsize_steps.py -q -x -s -k "config3 or steps" 2>&1 | grep "output nrmse\|guided\|passed\|failed" | tail -8
echo "== attention"; timeout 200 python tools/microbench.py --only attn 2>&1 | grep "^attention D=40"
for rep in 1 2 3; do timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('step', round(d['value'],3), round(d['ms_per_step'],2))"; done
} 2>&1 | tee gpurun_out/r04_s4.txt
