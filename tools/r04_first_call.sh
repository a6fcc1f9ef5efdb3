#!/bin/bash
# First GPU call of the next round (~14 GPU-minutes): settle the round-3 opt-in before building on it.
#   gpurun --timeout 1500 -- 'bash tools/r04_first_call.sh'
#  1. the whole -m gpu suite with the wide-tile GEMM kernel selected (HUMANVID_TUNING=3=4: N = 320, K >= 640, M % 256 == 0)
#  2. same-box step A/B, default selection against it, two repetitions
#  3. per-shape times of the GEMMs it takes and of the level-1 N = 640 shapes it does not take yet
# If 1 is green and 2 wins: make value 4's selection the default in hv_gemm_choose,
# re-run tools/final_r03.sh's measure-only leg, then build the 128-row form for N = 640.
mkdir -p gpurun_out
{
HUMANVID_TUNING=3=4 timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for rep in 1 2; do for m in 1 4; do HUMANVID_TUNING=3=$m timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('step glds=$m', d['value'], d['ms_per_step'])"; done; done
for m in 1 4; do HV_GEMM_GLDS=$m timeout 300 python tools/microbench.py --only gemm 2>&1 | grep "^gemm" | grep "N=320\|N=640" | awk -v v=$m '{printf "glds=%s %s\n", v, $0}'; done
} | tee gpurun_out/r04_first_call.txt
