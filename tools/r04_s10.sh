#!/bin/bash
# session 10: attention40 two-group loop -- GPU kernel tests, microbench (generic vs dedicated), step
mkdir -p gpurun_out
{
echo "== attention kernel tests"
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" 2>&1 | tail -4
echo "== attention microbench (QT=2 generic, QT=0 hv_attention40)"
timeout 300 python tools/microbench.py --only attention 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('step', d['value'], d['ms_per_step'])"
done
} > gpurun_out/r04_s10.txt 2>&1
