#!/bin/bash
# The round's closing measurement, all at the built tree that is shipped (run LAST, with >= 12 GPU-minutes left):
#   gpurun --timeout 2700 -- 'bash tools/final_r06.sh'
#  1. the -m gpu suite, serially          -> gpurun_out/r06_final_pytest.txt
#  2. the bench line (+ step profile)     -> gpurun_out/r06_bench_line_final.json, r06_step_profile_final.tsv
#  3. rocprofv3 --kernel-trace --stats    -> gpurun_out/r06_final_kernel_stats.csv
#  4. three counter passes, per shape     -> gpurun_out/r06_pmc_traffic.json
#     + the LDS bank-conflict pass        -> gpurun_out/r06_lds_conflicts.txt
# Copy 2-4 into profiles/ and commit them with the same tree.
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "$1" != "--no-tests" ] && [ "$1" != "--measure-only" ]; then
  timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r06_final_pytest.txt 2>&1
  tail -3 gpurun_out/r06_final_pytest.txt
fi
MEASURE_ONLY=0; if [ "$1" == "--measure-only" ]; then MEASURE_ONLY=1; fi   # bench line + the three traffic passes only
HV_PROFILE_DUMP=gpurun_out/r06_step_profile_final.tsv timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r06_bench_line_final.json 2> gpurun_out/r06_bench_final.err
tail -c 300 gpurun_out/r06_bench_final.err; cut -c1-400 gpurun_out/r06_bench_line_final.json
if [ $MEASURE_ONLY == 0 ]; then
{ timeout 300 python bench.py --config 2 --steps 8 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | grep '^{'
  timeout 400 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | grep '^{'
  timeout 400 python bench.py --config 5 --fp8-attention 1 --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | grep '^{'
} > gpurun_out/r06_bench_configs.jsonl; cut -c1-200 gpurun_out/r06_bench_configs.jsonl
bash tools/prof_bench.sh r06_final --no-profile | head -12
timeout 300 python tools/microbench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_microbench_final.txt
fi
REPO=$(pwd)
cd /tmp
CMD="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile"
PASSES=("fetch FETCH_SIZE" "write WRITE_SIZE" "mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE")
if [ $MEASURE_ONLY == 0 ]; then PASSES+=("lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"); fi
for pass in "${PASSES[@]}"; do
  set -- $pass; tag=$1; shift
  rm -rf /tmp/pmc_$tag
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$tag -- $CMD < /dev/null > $REPO/gpurun_out/r06_pmc_$tag.log 2>&1
done
cd $REPO
python tools/pmc_by_shape.py gpurun_out/r06_step_profile_final.tsv gpurun_out/r06_pmc_traffic.json /tmp/pmc_fetch /tmp/pmc_write /tmp/pmc_mfma 2>&1 | tail -22
# the bench line once more, now with roofline.traffic from the passes just collected (bench.py reads profiles/r06_pmc_traffic.json
# and refuses a file whose kernel-source digest is not this tree's)
cp gpurun_out/r06_pmc_traffic.json profiles/r06_pmc_traffic.json
# (default bench line: the CPU baseline is the WHOLE config-3 step on the host cores since round 5 -- ~5.5 min on 128 cores)
HV_PROFILE_DUMP=gpurun_out/r06_step_profile_final.tsv timeout 1200 python bench.py > gpurun_out/r06_bench_line_final.json 2> gpurun_out/r06_bench_final.err
python -c "
import json
d=json.loads(open('gpurun_out/r06_bench_line_final.json').read().strip().splitlines()[-1])
print('final line:', d['value'], 'steps/s', d['ms_per_step'], 'ms', {k: d['roofline'][k] for k in ('kernel', 'achieved', 'frac', 'traffic')})"
if [ $MEASURE_ONLY == 0 ]; then python tools/pmc_lds.py /tmp/pmc_lds > gpurun_out/r06_lds_conflicts.txt 2>&1; head -12 gpurun_out/r06_lds_conflicts.txt | cut -c1-160; fi
