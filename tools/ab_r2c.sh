#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/d_pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/d_pytest.txt
tail -8 gpurun_out/d_pytest.txt
HUMANVID_GN_PROLOGUE=1 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile > gpurun_out/d_bench_gnpro.txt 2>&1
HV_PROFILE_DUMP=gpurun_out/d_step_profile.tsv timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/d_bench.txt 2>&1
python - <<'PY'
import json
for f in ['gpurun_out/d_bench_gnpro.txt','gpurun_out/d_bench.txt']:
    for l in open(f):
        if l.startswith('{'):
            d=json.loads(l); print(f, d['value'], d['ms_per_step'])
            for k in d.get('kernels',[]): print('   ',k)
PY
