#!/bin/bash
# round 6, session 10: the CFG-parallel axis on one GPU: (a) the two-rank gloo test (bit-identical to the single-process run),
# (b) the step ONE rank of a two-rank CFG-parallel job runs (bench.py --cfg-half 0 / 1) beside the single-GPU step
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== tests/test_gpu_sharded.py -k cfg_parallel"
timeout 1200 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu -k "cfg_parallel" -s 2>&1 | grep -v "^$" | tail -6
for args in "" "--cfg-half 0" "--cfg-half 1"; do
echo "== bench.py $args"
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile $args 2>&1 | grep "^{\"metric" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],3), 'steps/s', round(d['ms_per_step'],2), 'ms |', d['config']['parallelism'][:120], '|', json.dumps(d.get('exchange'))[:300])"
done
} > gpurun_out/r06_s10_cfg_parallel.txt 2>&1
cat gpurun_out/r06_s10_cfg_parallel.txt
