#!/bin/bash
# rocprofv3 kernel-trace of the bench command, reduced to a per-kernel CSV: tools/prof_bench.sh <tag> [bench args...]
# Run on the GPU box from the repo root; writes gpurun_out/<tag>_kernel_stats.csv (copy the ones to keep into profiles/).
TAG=$1; shift
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" < /dev/null > $REPO/gpurun_out/${TAG}_prof.log 2>&1
cd $REPO
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
python tools/rocpd_stats.py "$DB" gpurun_out/${TAG}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline $* (MI355X, $TAG)" > /dev/null
head -14 gpurun_out/${TAG}_kernel_stats.csv
