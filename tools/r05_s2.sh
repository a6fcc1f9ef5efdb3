#!/bin/bash
# Round 5, session 2: timing-only ablations of hv_gemm_p8_kernel (HV_P8_ABL builds: results are wrong by construction, only the
# time is read) -- which resource bounds the k-loop?  1 no LDS-DMA in the loop, 2 no fragment reads, 4 no MFMAs, 8 every DMA
# piece from one hot 32 KiB, 32 no vmcnt waits.
mkdir -p gpurun_out
OUT=gpurun_out/r05_s2.txt
{
for v in base abl32 abl8 abl1 abl2 abl4 abl3 abl6 abl7; do
  echo "== variant $v"
  if [ $v == base ]; then LIBV=humanvid_amd/lib/libhumanvid_hip.so; else LIBV=tools/bin/lib_$v.so; fi
  HV_LIB=$LIBV HV_GEMM_P8=1 timeout 200 python tools/microbench.py --only gemm 2>&1 | grep "^gemm \(qkv\|ff1\)"
done
echo "== two-group loop (p8=0)"
HV_GEMM_P8=0 timeout 200 python tools/microbench.py --only gemm 2>&1 | grep "^gemm \(qkv\|ff1\)"
} > $OUT 2>&1
cat $OUT
