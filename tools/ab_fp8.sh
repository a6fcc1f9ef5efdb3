#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullwidth.py -x -q -k "fp8 or two_frames" -s 2>&1 | grep -v "^$" | tail -12
for f in 0 1; do timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --fp8-attention $f 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('fp8=$f', d['value'], d['ms_per_step'], d['dtype'])
        for k in d['kernels'][:4]: print('   ',k)"; done
timeout 600 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>&1 | tail -1 | cut -c1-400
