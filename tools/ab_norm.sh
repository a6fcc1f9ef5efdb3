#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "layernorm or groupnorm or lnfold or fused" 2>&1 | tail -2
for L in tools/bin/lib_prevnorm.so humanvid_amd/lib/libhumanvid_hip.so; do
  n=$(basename $L .so)
  HV_LIB=$L timeout 200 python tools/microbench.py --only norm,gemm 2>&1 | grep -i "norm" | sed "s/^/$n /"
  HUMANVID_HIP_LIB=$L timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$n', d['value'], d['ms_per_step'])
        for k in d['kernels']:
            if 'gn_' in k['kernel'] or 'ln_' in k['kernel']: print('   ',k)"
done
