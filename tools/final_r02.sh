#!/bin/bash
# Round-end evidence in one gpurun call: full -m gpu suite (values printed), default bench line (with CPU baseline),
# rocprofv3 kernel-trace summary, PMC passes (HBM traffic + MFMA busy) -> gpurun_out/final_*
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -q -m gpu -s ) > gpurun_out/final_pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/final_pytest.txt
grep -E "passed|failed|error" gpurun_out/final_pytest.txt | tail -3
timeout 600 python bench.py > gpurun_out/final_bench.txt 2> gpurun_out/final_bench.err; tail -c 600 gpurun_out/final_bench.txt | head -c 300; echo
HV_PROFILE_DUMP=gpurun_out/final_step_profile.tsv timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/final_bench2.txt 2>&1
bash tools/prof_bench.sh final_r02 --no-profile > gpurun_out/final_prof_head.txt 2>&1; head -8 gpurun_out/final_prof_head.txt
bash tools/pmc_passes.sh > gpurun_out/final_pmc_tail.txt 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_summary.csv gpurun_out/final_pmc_traffic.json "$(cat gpurun_out/../.git_commit 2>/dev/null || echo r02-final)" | cut -c1-400
timeout 300 python bench.py --config 2 --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | tail -1 | cut -c1-250
timeout 600 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>&1 | tail -1 | cut -c1-250
