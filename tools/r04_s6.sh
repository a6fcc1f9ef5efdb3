#!/bin/bash
# Round 4, GPU call 6: phase timeline (s_memtime) of the ping-pong k-loop next to the lockstep loop at three shapes.
mkdir -p gpurun_out
{
for args in "294912 2560 320 2 1" "73728 5120 640 2 1" "18432 10240 1280 2 1"; do
  tools/bin/gemm_trace $args 0 | tail -3
  tools/bin/gemm_trace $args 1 | tail -4
done
} 2>&1 | cut -c1-600 | tee gpurun_out/r04_s6.txt
