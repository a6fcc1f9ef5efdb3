#!/bin/bash
# Same-box A/B of spatial-attention build variants: tools/r03_attn_ab.sh lib_a.so lib_b.so ...  (microbench at the config-#3
# shapes for each, then the attention kernel tests on the LAST one)
mkdir -p gpurun_out
for L in "$@"; do
  n=$(basename $L .so)
  HV_LIB=$L timeout 300 python tools/microbench.py --only attn > gpurun_out/r03_attn_$n.txt 2>&1
  echo "== $n"; grep -i "attention" gpurun_out/r03_attn_$n.txt | grep -v "QT=4\|QT=1"
done
last="${@: -1}"
HUMANVID_HIP_LIB=$last timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k attention 2>&1 | tail -3
