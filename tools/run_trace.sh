#!/bin/bash
mkdir -p gpurun_out
{
for args in "294912 960 320 1 9" "294912 960 320 1 6" "294912 960 320 1 2" "294912 2560 320 2 9" "294912 320 320 3 9" "294912 320 1280 3 9" "73728 5120 640 2 9" "18432 10240 1280 2 9" "18432 1280 1280 3 9"; do
  tools/bin/gemm_trace $args
done
} > gpurun_out/gemm_trace_r2.txt 2>&1
cat gpurun_out/gemm_trace_r2.txt
