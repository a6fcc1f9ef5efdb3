#!/bin/bash
# one gpurun call: kernel parity on the new library, then same-box A/B of the previous and the new library
# (microbench per kernel family + the bench step).  Output: gpurun_out/ab_*.txt
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q > gpurun_out/ab_pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/ab_pytest.txt
tail -5 gpurun_out/ab_pytest.txt
for L in tools/bin/lib_base.so humanvid_amd/lib/libhumanvid_hip.so; do
  n=$(basename $L .so)
  HV_LIB=$L timeout 300 python tools/microbench.py --only gemm,conv > gpurun_out/ab_micro_$n.txt 2>&1
  HUMANVID_HIP_LIB=$L timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/ab_bench_$n.txt 2>&1
  tail -c 600 gpurun_out/ab_bench_$n.txt | head -c 300; echo
done
python - <<'PY'
import re
def rd(f):
    d={}
    for l in open(f):
        m=re.match(r'(.{58})\s+([\d.]+) ms',l)
        if m: d[m.group(1).strip()]=float(m.group(2))
    return d
a=rd('gpurun_out/ab_micro_lib_base.txt'); b=rd('gpurun_out/ab_micro_libhumanvid_hip.txt')
for k in a:
    if k in b: print('%-58s %8.3f -> %8.3f  x%.2f'%(k,a[k],b[k],a[k]/b[k]))
PY
