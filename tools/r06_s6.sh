#!/bin/bash
# round 6, session 6: the four-wave kernel as shipped (default: deferred forms at K >= 640): GPU kernel tests incl. the bit-wise
# bench-shape comparison, then the step (bench.py) against HUMANVID_TUNING=10=0, interleaved
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== kernel tests (gemm)"
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" 2>&1 | tail -3
for rep in 1 2; do
for t in "" "10=0"; do
echo "== bench HUMANVID_TUNING=$t"
HUMANVID_TUNING="$t" timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
done
} > gpurun_out/r06_s6.txt 2>&1
cat gpurun_out/r06_s6.txt
