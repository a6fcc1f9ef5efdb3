#!/bin/bash
# Same-box A/B of the conv epilogue's store order: column pass by column pass (base) vs deferred row-major stores
mkdir -p gpurun_out
V="tools/bin/lib_conv_base.so humanvid_amd/lib/libhumanvid_hip.so"
{
for v in $V; do HV_LIB=$v timeout 300 python tools/microbench.py --only conv 2>&1 | grep "conv3x3" | awk -v v=$(basename $v) '{printf "%-22s %s\n", v, $0}'; done
HUMANVID_HIP_LIB=humanvid_amd/lib/libhumanvid_hip.so timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv or gn_parts" 2>&1 | tail -1
for rep in 1 2; do for v in $V; do HUMANVID_HIP_LIB=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('step $v', d['value'], d['ms_per_step'])"; done; done
} | tee gpurun_out/r03_conv_stores_ab.txt
