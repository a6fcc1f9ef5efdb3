#!/bin/bash
# round 6, session 11: per-kernel time of the two half steps of the CFG-parallel axis (rocprofv3 --kernel-trace --stats)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
REPO=$(pwd)
export TMPDIR=/tmp
for h in 0 1; do
  cd /tmp; rm -rf /tmp/prof_half$h
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_half$h -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --cfg-half $h < /dev/null > $REPO/gpurun_out/r06_s11_half$h.log 2>&1
  cd $REPO
  DB=$(find /tmp/prof_half$h -name "*.db" | head -1)
  python tools/rocpd_stats.py "$DB" gpurun_out/r06_s11_half${h}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --cfg-half $h (MI355X)" > /dev/null
  head -16 gpurun_out/r06_s11_half${h}_kernel_stats.csv | cut -c1-150
done
