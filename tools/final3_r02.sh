#!/bin/bash
# Round-end evidence after the tile-policy-10 / wave-merge GroupNorm finalize change, in one gpurun call:
# full -m gpu suite (three xdist workers: the suite is host-bound in the fp32 oracle), default bench line, launch profile,
# rocprofv3 kernel-trace summary, then a same-box A/B of the step against the previous library.
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -q -m gpu -n 3 --dist loadfile ) > gpurun_out/final3_pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/final3_pytest.txt
grep -E "passed|failed|error" gpurun_out/final3_pytest.txt | tail -3
if ! grep -q "pytest rc=0" gpurun_out/final3_pytest.txt; then
  ( time timeout 600 python -m pytest tests -q -m gpu --lf -x -s ) > gpurun_out/final3_pytest_lf.txt 2>&1; echo "pytest-lf rc=$?" >> gpurun_out/final3_pytest_lf.txt
  grep -E "passed|failed|error|rc=" gpurun_out/final3_pytest_lf.txt | tail -3
fi
timeout 400 python bench.py > gpurun_out/final3_bench.txt 2> gpurun_out/final3_bench.err; python - <<'PY'
import json
for l in open('gpurun_out/final3_bench.txt'):
    if l.startswith('{'):
        d = json.loads(l); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])
PY
HV_PROFILE_DUMP=gpurun_out/final3_step_profile.tsv timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/final3_bench2.txt 2>&1
for L in tools/bin/lib_prev_r2e.so humanvid_amd/lib/libhumanvid_hip.so; do
  HUMANVID_HIP_LIB=$L timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$L', d['value'], d['ms_per_step'])"
done
bash tools/prof_bench.sh final3_r02 --no-profile > gpurun_out/final3_prof_head.txt 2>&1; head -14 gpurun_out/final3_prof_head.txt
