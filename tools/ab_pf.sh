#!/bin/bash
mkdir -p gpurun_out
for m in 0 1 2 3 4; do
  HV_GEMM_PF=$m timeout 200 python tools/microbench.py --only gemm > gpurun_out/pf_$m.txt 2>&1
done
python - <<'PY'
import re
def rd(f):
    d={}
    for l in open(f):
        m=re.match(r'(.{58})\s+([\d.]+) ms',l)
        if m: d[m.group(1).strip()]=float(m.group(2))
    return d
ms=[0,1,2,3,4]
D={m:rd('gpurun_out/pf_%d.txt'%m) for m in ms}
print('%-50s'%'shape'+''.join('%8s'%('pf%d'%m) for m in ms))
for k in D[0]:
    print('%-50s'%k[:50]+''.join('%8.3f'%D[m].get(k,float('nan')) for m in ms))
PY
HV_GEMM_PF=2 timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" 2>&1 | tail -2
