#!/bin/bash
# Round 5, session 11: the -m gpu suite and the bench line on the final tree (kernel sources unchanged since the closing
# measurement: same digest as profiles/r05_pmc_traffic.json; Python-side changes since: the one-graph sharded replay).
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r05_final2_pytest.txt 2>&1
tail -3 gpurun_out/r05_final2_pytest.txt
timeout 400 python bench.py --no-cpu-baseline 2>gpurun_out/r05_final2_bench.err | grep "^{" > gpurun_out/r05_final2_bench_line.json
python -c "
import json
d=json.loads(open('gpurun_out/r05_final2_bench_line.json').read().strip().splitlines()[-1])
print('line:', d['value'], 'steps/s', d['ms_per_step'], 'ms', {k: d['roofline'][k] for k in ('kernel', 'achieved', 'frac', 'traffic')})"
