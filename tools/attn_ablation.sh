#!/bin/bash
# Ablation builds of the spatial-attention kernel (timing only, results are wrong): which parts of a 64-key tile cost what.
#   bash tools/attn_ablation.sh build      (here: hipcc)      bash tools/attn_ablation.sh run   (on the GPU box)
BITS="0 1 2 4 8 16 32 48 63"
if [ "$1" == "build" ]; then
  for b in $BITS; do bash tools/build_variant.sh attn_abl$b k_attention -DHV_ATTN_ABL=$b > /dev/null; done
  ls tools/bin/lib_attn_abl*.so
else
  mkdir -p gpurun_out
  for b in $BITS; do
    printf "ABL=%-3s " $b
    HV_LIB=tools/bin/lib_attn_abl$b.so timeout 200 python tools/microbench.py --only attn 2>&1 | grep "attention D=40" | grep -v "QT=4" | awk '{print $8, $9, $10, $11}'
  done | tee gpurun_out/r03_attn_ablation.txt
fi
