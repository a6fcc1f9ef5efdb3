#!/bin/bash
# Round 5, session 12: how much of a GEMM launch is the exposed epilogue?  Timing-only builds of the shipped LDS-DMA kernels
# (scratch copies of csrc/, tools/bin/src_abl): "noepi" = k-loop only (the accumulators stay live, nothing is converted or
# stored), "noloop" = epilogue + barriers only (no DMA, no fragment reads, no MFMAs: the epilogue converts zeros).
# Build (here, hipcc cross-compiles): S=tools/bin/src_abl; mkdir -p $S; cp humanvid_amd/csrc/*.h humanvid_amd/csrc/k_gemm.hip $S/;
#   patch $S/hv_gemm.h < profiles/r05_gemm_ablation_hooks.patch; for v in NOEPI NOLOOP: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC
#   -Iinclude -I$S -DHV_ABL_$v -x hip -c $S/k_gemm.hip -o k_gemm.o; link it with the other objects of humanvid_amd/lib/obj/
#   into tools/bin/lib_gemm_noepi.so / lib_gemm_noloop.so (as tools/build_variant.sh does).
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
