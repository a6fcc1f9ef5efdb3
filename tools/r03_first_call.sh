#!/bin/bash
# First GPU call of round 3 (prepared at the end of round 2, when the GPU budget was spent): hardware check + timing of the
# variants written blind -- GEMM tile policies 10 (fill test), 11 / 13 (two LDS-DMA readiness groups under counted vmcnt,
# 13 at the eight-phase cadence, 15 / 16 in the 128x128x64 kernel as well), convolution with 64-channel chunks (HV_TUNE_CONV_BIG = 3) -- then the step on one box
# under the combinations.  ~7 GPU-minutes:   gpurun --timeout 560 -- 'bash tools/r03_first_call.sh'
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 150 tools/bin/hwcheck --experimental --bench > gpurun_out/r03_hwcheck.txt 2>&1
echo "hwcheck rc=$?" >> gpurun_out/r03_hwcheck.txt
tail -45 gpurun_out/r03_hwcheck.txt
: > gpurun_out/r03_step_ab.txt
for cfg in "9 1" "10 1" "11 1" "13 1" "12 1" "14 1" "15 1" "16 1" "9 3" "12 3" "9 1"; do
  set -- $cfg
  HUMANVID_GEMM_GLDS=$1 HUMANVID_CONV_BIG=$2 timeout 240 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('gemm policy $1  conv $2 : %.3f steps/s  %.2f ms' % (d['value'], d['ms_per_step']))" | tee -a gpurun_out/r03_step_ab.txt
done
