#!/bin/bash
mkdir -p gpurun_out
for m in 1 2; do
  HV_CONV_BIG=$m timeout 200 python tools/microbench.py --only conv > gpurun_out/cv_$m.txt 2>&1
done
python - <<'PY'
import re
def rd(f):
    d={}
    for l in open(f):
        m=re.match(r'(.{58})\s+([\d.]+) ms',l)
        if m: d[m.group(1).strip()]=float(m.group(2))
    return d
a=rd('gpurun_out/cv_1.txt'); b=rd('gpurun_out/cv_2.txt')
for k in a: print('%-58s %8.3f -> %8.3f  x%.2f'%(k,a[k],b.get(k,0),a[k]/b[k] if b.get(k) else 0))
PY
HV_CONV_BIG=2 timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv" 2>&1 | tail -2
for m in 1 2; do HUMANVID_CONV_BIG=$m timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('conv_big $m', d['value'], d['ms_per_step'])"; done
