#!/bin/bash
# Round 4, GPU call 5: the ping-pong k-loop of the 256 x 256 x 64 GEMM tiles (correctness on the hardware: bit-identity against
# the lockstep loop, repeated; then per-shape times and the step), and the attention overflow test after its robustness fix.
#   gpurun --timeout 900 -- 'bash tools/r04_s5.sh'
mkdir -p gpurun_out
{
echo "== kernel tests"; timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "ping_pong or forced_rescale or attention_variants" 2>&1 | tail -15
for pp in 0 1; do HUMANVID_TUNING=8=$pp timeout 300 python tools/microbench.py --only gemm 2>&1 | grep "^gemm" | grep "qkv\|ff1" | awk -v v=$pp '{printf "pp=%s %s\n", v, $0}'; done
for rep in 1 2; do for tune in "8=0" "8=1"; do HUMANVID_TUNING=$tune timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('step tuning=$tune', round(d['value'],3), round(d['ms_per_step'],2))"; done; done
echo "== full-size forward parity under the ping-pong loop"; HUMANVID_TUNING=8=1 timeout 400 python -m pytest tests/test_gpu_fullsize_parity.py -q -x -s -k "config3" 2>&1 | grep "output nrmse\|passed\|failed"
} 2>&1 | tee gpurun_out/r04_s5.txt
