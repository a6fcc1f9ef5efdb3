#!/bin/bash
# A/B of the stride-1 conv tile shapes: tuning key 5 = 1 (128-pixel tiles), 4 (256 pixels on 8 waves), 5 (256 pixels on 4 waves)
mkdir -p gpurun_out
for m in 1 4 5; do HV_CONV_BIG=$m timeout 300 python tools/microbench.py --only conv > gpurun_out/cvt_$m.txt 2>&1; done
python - <<'PY' | tee gpurun_out/r03_conv_tiles_ab.txt
import re
V="1 4 5".split()
def rd(f):
    d={}
    for l in open(f):
        m=re.match(r'(.{58})\s+([\d.]+) ms',l)
        if m: d[m.group(1).strip()]=float(m.group(2))
    return d
r=[rd('gpurun_out/cvt_%s.txt'%v) for v in V]
print('%-58s'%'case (ms)'+''.join('%12s'%v for v in V))
for k in r[0]: print('%-58s'%k+''.join('%12.3f'%x.get(k,0) for x in r))
PY
for m in 4 5; do HUMANVID_TUNING=5=$m timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv" 2>&1 | tail -1; done | tee -a gpurun_out/r03_conv_tiles_ab.txt
for m in 1 4 5; do HUMANVID_TUNING=5=$m timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('step conv_big=$m', d['value'], d['ms_per_step'])"; done | tee -a gpurun_out/r03_conv_tiles_ab.txt
