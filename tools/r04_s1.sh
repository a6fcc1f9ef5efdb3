#!/bin/bash
# Round 4, GPU call 1 (~14 GPU-minutes): the dedicated head-dim-40 attention kernel (hv_attention40.h), the spill-free epilogue
# of the wide-tile GEMM kernel as the default for N = 320 / K >= 640, and the loop-body parity fixtures at the benchmarked size.
#   gpurun --timeout 1500 -- 'bash tools/r04_s1.sh'
mkdir -p gpurun_out
{
echo "== kernel tests"; timeout 700 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention or gemm or parts or statistics" 2>&1 | tail -4
echo "== full-size forward parity (config 3)"; timeout 400 python -m pytest tests/test_gpu_fullsize_parity.py -q -x -s -k "config3" 2>&1 | grep -v "^\[config3\] .*slice nrmse" | tail -4
echo "== loop-body parity at full size"; timeout 600 python -m pytest tests/test_gpu_fullsize_steps.py -q -x -s 2>&1 | grep "nrmse\|passed\|failed\|Error\|error" | tail -12
echo "== CFG halves on two streams (2 ranks on one GPU, host-staged)"; timeout 400 python -m pytest tests/test_gpu_sharded.py -q -x -k "cfg_halves or matches_oracle" 2>&1 | tail -3
echo "== attention microbench (variants 2 = generic, 1 = attention40 head-major, 0 = attention40)"; timeout 300 python tools/microbench.py --only attn 2>&1 | grep "^attention"
for m in 6 1 4 5; do HV_GEMM_GLDS=$m timeout 300 python tools/microbench.py --only gemm 2>&1 | grep "^gemm" | grep "N=320 \|N=640 \|LN fold" | awk -v v=$m '{printf "glds=%s %s\n", v, $0}'; done
for rep in 1 2; do for tune in "0=2,3=6" "0=0,3=6" "0=1,3=6" "0=0,3=1" "0=0,3=4"; do HUMANVID_TUNING=$tune timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('step tuning=$tune', round(d['value'],3), round(d['ms_per_step'],2))"; done; done
} 2>&1 | tee gpurun_out/r04_s1.txt
