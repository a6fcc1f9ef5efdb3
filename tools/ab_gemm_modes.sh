#!/bin/bash
mkdir -p gpurun_out
for m in 2 1 3 4 6 7 8; do
  HV_GEMM_GLDS=$m timeout 200 python tools/microbench.py --only gemm > gpurun_out/gm_$m.txt 2>&1
done
python - <<'PY'
import re
def rd(f):
    d={}
    for l in open(f):
        m=re.match(r'(.{58})\s+([\d.]+) ms',l)
        if m: d[m.group(1).strip()]=float(m.group(2))
    return d
ms=[2,1,3,4,6,7,8]
D={m:rd('gpurun_out/gm_%d.txt'%m) for m in ms}
print('%-50s'%'shape'+''.join('%8s'%('m%d'%m) for m in ms))
for k in D[2]:
    print('%-50s'%k[:50]+''.join('%8.3f'%D[m].get(k,float('nan')) for m in ms))
PY
