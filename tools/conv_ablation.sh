#!/bin/bash
# Ablation builds of the 3x3 convolution kernel (timing only, results are wrong): which parts of a tile cost what.
#   bash tools/conv_ablation.sh build      (here: hipcc)      bash tools/conv_ablation.sh run   (on the GPU box)
BITS="0 1 2 3 4 8 16 32 7 24 63"
if [ "$1" == "build" ]; then
  for b in $BITS; do bash tools/build_variant.sh conv_abl$b k_conv -DHV_CONV_ABL=$b > /dev/null; done
  ls tools/bin/lib_conv_abl*.so
else
  mkdir -p gpurun_out
  for b in $BITS; do
    HV_LIB=tools/bin/lib_conv_abl$b.so timeout 200 python tools/microbench.py --only conv 2>&1 | grep "conv3x3 plain\|upsample\|stride-2" | awk -v b=$b '{printf "ABL=%-3s %s\n", b, $0}'
  done | tee gpurun_out/r03_conv_ablation.txt
fi
