#!/bin/bash
# Round 5, session 3: hv_gemm_p8_kernel with scalar-base addressing; SCHED 0 (8 intervals, key 8 = 1) and SCHED 1 (one barrier per
# k-tile, key 8 = 2) against the two-group loop (key 8 = 0); HV_P8_PRIO builds: 0 none, 1 around the MFMA groups, 2 static for waves 4-7.
mkdir -p gpurun_out
OUT=gpurun_out/r05_s3.txt
{
echo "== bit-identity at the bench shapes"
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "eight_interval" 2>&1 | tail -3
for p8 in 0 1 2; do
  echo "== microbench p8=$p8 (PRIO 1)"
  HV_GEMM_P8=$p8 timeout 200 python tools/microbench.py --only gemm 2>&1 | grep "^gemm \(qkv\|ff1\)"
done
for v in prio2 prio0; do
  echo "== microbench p8=2 lib_$v"
  HV_LIB=tools/bin/lib_$v.so HV_GEMM_P8=2 timeout 200 python tools/microbench.py --only gemm 2>&1 | grep "^gemm \(qkv\|ff1\)"
done
} > $OUT 2>&1
cat $OUT
