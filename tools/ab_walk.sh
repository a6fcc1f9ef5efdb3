#!/bin/bash
mkdir -p gpurun_out
for cfg in "0 humanvid_amd/lib/libhumanvid_hip.so" "1 humanvid_amd/lib/libhumanvid_hip.so" "1 tools/bin/lib_gemm_sc1.so" "0 tools/bin/lib_gemm_sc1.so"; do
  set -- $cfg
  n=walk$1_$(basename $2 .so)
  HV_LIB=$2 HV_GEMM_WALK=$1 timeout 200 python tools/microbench.py --only gemm > gpurun_out/w_$n.txt 2>&1
done
python - <<'PY'
import re,glob
def rd(f):
    d={}
    for l in open(f):
        m=re.match(r'(.{58})\s+([\d.]+) ms',l)
        if m: d[m.group(1).strip()]=float(m.group(2))
    return d
fs=sorted(glob.glob('gpurun_out/w_*.txt'))
D=[rd(f) for f in fs]
print(' | '.join(f[13:-4] for f in fs))
for k in D[0]:
    print('%-50s'%k[:50]+''.join('%8.3f'%d.get(k,float('nan')) for d in D))
PY
for w in 0 1; do HUMANVID_GEMM_WALK=$w timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('walk $w', d['value'], d['ms_per_step'])"; done
HUMANVID_HIP_LIB=tools/bin/lib_gemm_sc1.so HUMANVID_GEMM_WALK=1 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('walk 1 sc1', d['value'], d['ms_per_step'])"
for args in "294912 960 320 1 9" "294912 2560 320 2 9" "294912 320 320 3 9"; do tools/bin/gemm_trace $args | tail -3; done
