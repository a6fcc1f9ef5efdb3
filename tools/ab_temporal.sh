#!/bin/bash
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "temporal" 2>&1 | tail -2
for L in tools/bin/lib_prevtemporal.so humanvid_amd/lib/libhumanvid_hip.so; do
  n=$(basename $L .so)
  HV_LIB=$L timeout 200 python tools/microbench.py --only temporal 2>&1 | grep -i "temporal" | sed "s/^/$n /"
done
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | tail -1 | cut -c1-200
