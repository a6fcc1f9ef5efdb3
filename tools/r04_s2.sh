#!/bin/bash
# Round 4, GPU call 2 (~12 GPU-minutes): phase stagger of the persistent GEMM workgroups, the rewritten GroupNorm-apply pass, the
# convolution's patch-major raster, and the CFG-halves overlap test with its full output.
#   gpurun --timeout 1300 -- 'bash tools/r04_s2.sh'
mkdir -p gpurun_out
{
echo "== kernel tests"; timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm or affine or conv or parts" 2>&1 | tail -3
echo "== CFG halves on two streams"; timeout 400 python -m pytest tests/test_gpu_sharded.py -q -x -k "cfg_halves" 2>&1 | tail -40
echo "== microbench: stagger"; for m in 0 2 4 8; do HUMANVID_TUNING=8=$m timeout 300 python tools/microbench.py --only gemm 2>&1 | grep "^gemm" | awk -v v=$m '{printf "stagger=%s %s\n", v, $0}'; done
echo "== microbench: conv raster"; for m in 0 1; do HUMANVID_TUNING=9=$m timeout 300 python tools/microbench.py --only conv 2>&1 | grep "^conv" | grep plain | awk -v v=$m '{printf "raster=%s %s\n", v, $0}'; done
for rep in 1 2; do for tune in "8=0" "8=2" "8=4" "8=8" "9=2" "8=4,9=2"; do HUMANVID_TUNING=$tune timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('step tuning=$tune', round(d['value'],3), round(d['ms_per_step'],2))"; done; done
HV_PROFILE_DUMP=gpurun_out/r04_s2_step_profile.tsv timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-300
grep "affine\|from_parts" gpurun_out/r04_s2_step_profile.tsv | head -40
} 2>&1 | tee gpurun_out/r04_s2.txt
