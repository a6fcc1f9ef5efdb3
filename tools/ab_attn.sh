#!/bin/bash
mkdir -p gpurun_out
for L in "$@"; do
  n=$(basename $L .so)
  HV_LIB=$L timeout 300 python tools/microbench.py --only attn > gpurun_out/c_attn_$n.txt 2>&1
  echo "== $n"; grep "attention" gpurun_out/c_attn_$n.txt | grep -v "QT=4\|QT=1"
done
HUMANVID_HIP_LIB=$1 timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k attention 2>&1 | tail -2
