#!/bin/bash
# A/B of the conv kernel's LDS layouts on one box (build the variants here first):
#   bash tools/build_variant.sh conv_old k_conv -DHV_CONV_PITCH32=0
#   bash tools/build_variant.sh conv_p32 k_conv
#   bash tools/build_variant.sh conv_p32h1 k_conv -DHV_CONV_HBUFS1=1
mkdir -p gpurun_out
V="conv_old conv_p32 conv_p32h1"
for v in $V; do HV_LIB=tools/bin/lib_$v.so timeout 300 python tools/microbench.py --only conv > gpurun_out/cvl_$v.txt 2>&1; done
python - <<'PY' | tee gpurun_out/r03_conv_lds_ab.txt
import re
V="conv_old conv_p32 conv_p32h1".split()
def rd(f):
    d={}
    for l in open(f):
        m=re.match(r'(.{58})\s+([\d.]+) ms',l)
        if m: d[m.group(1).strip()]=float(m.group(2))
    return d
r=[rd('gpurun_out/cvl_%s.txt'%v) for v in V]
print('%-58s'%'case (ms)'+''.join('%12s'%v for v in V))
for k in r[0]: print('%-58s'%k+''.join('%12.3f'%x.get(k,0) for x in r))
PY
for v in $V; do HUMANVID_HIP_LIB=tools/bin/lib_$v.so timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv" 2>&1 | tail -1; done | tee -a gpurun_out/r03_conv_lds_ab.txt
for rep in 1 2; do for v in $V; do HUMANVID_HIP_LIB=tools/bin/lib_$v.so timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('step $v', d['value'], d['ms_per_step'])"; done; done | tee -a gpurun_out/r03_conv_lds_ab.txt
