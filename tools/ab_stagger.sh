#!/bin/bash
mkdir -p gpurun_out
for m in 0 8 16 24 32 48 64 96; do
  HV_GEMM_STAGGER=$m timeout 200 python tools/microbench.py --only gemm > gpurun_out/st_$m.txt 2>&1
done
python - <<'PY'
import re
def rd(f):
    d={}
    for l in open(f):
        m=re.match(r'(.{58})\s+([\d.]+) ms',l)
        if m: d[m.group(1).strip()]=float(m.group(2))
    return d
ms=[0,8,16,24,32,48,64,96]
D={m:rd('gpurun_out/st_%d.txt'%m) for m in ms}
print('%-50s'%'shape'+''.join('%8s'%('s%d'%m) for m in ms))
for k in D[0]:
    print('%-50s'%k[:50]+''.join('%8.3f'%D[m].get(k,float('nan')) for m in ms))
PY
for m in 0 16 32 64; do HUMANVID_GEMM_STAGGER=$m timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('stagger $m', d['value'], d['ms_per_step'])"; done
