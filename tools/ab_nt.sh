#!/bin/bash
mkdir -p gpurun_out
for L in humanvid_amd/lib/libhumanvid_hip.so tools/bin/lib_gemm_nt.so; do
  n=$(basename $L .so)
  HV_LIB=$L timeout 200 python tools/microbench.py --only gemm > gpurun_out/nt_$n.txt 2>&1
done
python - <<'PY'
import re
def rd(f):
    d={}
    for l in open(f):
        m=re.match(r'(.{58})\s+([\d.]+) ms',l)
        if m: d[m.group(1).strip()]=float(m.group(2))
    return d
a=rd('gpurun_out/nt_libhumanvid_hip.txt'); b=rd('gpurun_out/nt_lib_gemm_nt.txt')
for k in a: print('%-58s %8.3f -> %8.3f  x%.2f'%(k,a[k],b.get(k,0),a[k]/b[k] if b.get(k) else 0))
PY
