#!/bin/bash
# Round 4, GPU call 8: where inside the GEGLU epilogue's "store issue" the cycles go (s_memtime marks around the wait and the stores).
mkdir -p gpurun_out
{
for args in "294912 2560 320 2 1" "18432 10240 1280 2 1"; do tools/bin/gemm_trace $args | tail -4; done
} 2>&1 | cut -c1-500 | tee gpurun_out/r04_s8.txt
