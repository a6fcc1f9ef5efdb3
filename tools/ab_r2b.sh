#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q > gpurun_out/b_pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/b_pytest.txt
tail -4 gpurun_out/b_pytest.txt
for L in tools/bin/lib_attn_nodefer.so humanvid_amd/lib/libhumanvid_hip.so tools/bin/lib_attn_thr4.so tools/bin/lib_attn_occ3.so; do
  n=$(basename $L .so)
  HV_LIB=$L timeout 300 python tools/microbench.py --only attn > gpurun_out/b_attn_$n.txt 2>&1
  echo "== $n"; grep attention gpurun_out/b_attn_$n.txt
done
HV_PROFILE_DUMP=gpurun_out/b_step_profile.tsv timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/b_bench.txt 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/b_bench.txt'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'])
        for k in d['kernels']: print('   ',k)
PY
