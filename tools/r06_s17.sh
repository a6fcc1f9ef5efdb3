#!/bin/bash
# round 6, session 17: per-rank kernel time of the two multi-GPU decompositions on one GPU (no exchange: the clip is cut to the
# rank's local frame count, as profiles/r05_s6 did): N-way frame shard (B = 2, 24 / N frames) vs CFG-parallel x N/2-way frame
# shard (B = 1 of one half, 48 / N frames)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for f in 24 12 6 3; do
echo "== frame shard: B = 2, $f frames (N = $((24 / f)))"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --frames $f 2>&1 | grep "^{\"metric" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms')"
done
for f in 24 12 6; do
for h in 0 1; do
echo "== CFG-parallel: half $h, B = 1, $f frames (N = $((48 / f)))"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --frames $f --cfg-half $h 2>&1 | grep "^{\"metric" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms')"
done
done
} > gpurun_out/r06_s17_per_rank.txt 2>&1
cat gpurun_out/r06_s17_per_rank.txt
