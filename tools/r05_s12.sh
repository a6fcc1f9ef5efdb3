#!/bin/bash
# Round 5, session 12: how much of a GEMM launch is the exposed epilogue?  Timing-only builds of the shipped LDS-DMA kernels
# (scratch copies of csrc/, tools/bin/src_abl): "noepi" = k-loop only (the accumulators stay live, nothing is converted or
# stored), "noloop" = epilogue + barriers only (no DMA, no fragment reads, no MFMAs: the epilogue converts zeros).
mkdir -p gpurun_out
OUT=gpurun_out/r05_s12.txt
{
for i in 1 2; do
for v in "" tools/bin/lib_gemm_noepi.so tools/bin/lib_gemm_noloop.so; do
echo "== microbench gemm: ${v:-shipped library}"
HV_LIB=$v timeout 300 python tools/microbench.py --only gemm 2>&1 | grep "gemm " | head -24
done
done
} > $OUT 2>&1
cat $OUT
