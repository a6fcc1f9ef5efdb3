#!/bin/bash
# Diagnosis builds of the temporal-attention kernel (see HV_TEMPORAL_FENCE / HV_TEMPORAL_TAIL in hv_temporal.h):
# tools/bin/lib_f<fence>t<tail>.so, consumed by tools/diag_fence.py on the GPU box.
set -e
cd "$(dirname "$0")/.."
for v in "0 0" "0 1" "0 2" "0 3" "2 0" "3 0" "4 0" "5 0" "6 0" "1 0"; do
  set -- $v
  tools/build_variant.sh f$1t$2 k_temporal -DHV_TEMPORAL_FENCE=$1 -DHV_TEMPORAL_TAIL=$2 &
done
wait
