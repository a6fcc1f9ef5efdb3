#!/bin/bash
mkdir -p gpurun_out
for L in humanvid_amd/lib/libhumanvid_hip.so tools/bin/lib_gemm_nostore.so tools/bin/lib_gemm_noepi.so tools/bin/lib_gemm_nodma.so; do
  n=$(basename $L .so)
  HV_LIB=$L timeout 200 python tools/microbench.py --only gemm > gpurun_out/dbg_$n.txt 2>&1
done
python - <<'PY'
import re,glob
def rd(f):
    d={}
    for l in open(f):
        m=re.match(r'(.{58})\s+([\d.]+) ms',l)
        if m: d[m.group(1).strip()]=float(m.group(2))
    return d
names=['libhumanvid_hip','lib_gemm_nostore','lib_gemm_noepi','lib_gemm_nodma']
D=[rd('gpurun_out/dbg_%s.txt'%n) for n in names]
print('%-50s'%'shape'+''.join('%10s'%n[-8:] for n in names))
for k in D[0]:
    print('%-50s'%k[:50]+''.join('%10.3f'%d.get(k,float('nan')) for d in D))
PY
