#!/bin/bash
# Phase timeline of the LDS-DMA GEMM at the step's hottest shapes (build tools/bin/gemm_trace first: see tools/gemm_trace.hip)
mkdir -p gpurun_out
{
for args in "294912 2560 320 2 1" "294912 960 320 1 1" "294912 320 1280 3 1" "294912 320 320 3 1" "73728 5120 640 2 1" "73728 640 2560 3 1" "18432 10240 1280 2 1" "18432 1280 5120 3 1"; do
  tools/bin/gemm_trace $args
done
} > gpurun_out/r04_gemm_trace.txt 2>&1
cut -c1-400 gpurun_out/r04_gemm_trace.txt
