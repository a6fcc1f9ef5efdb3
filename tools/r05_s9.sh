#!/bin/bash
# Round 5, session 9: the recorded sharded step (segments + RCCL collectives) as ONE captured graph (HUMANVID_STEP_GRAPH=1):
# bit-identity on a one-rank RCCL group, then the per-rank step time of the sharded code path with and without it.
mkdir -p gpurun_out
OUT=gpurun_out/r05_s9.txt
{
echo "== bit-identity: step graph vs segment replay (one-rank RCCL group, serial replay)"
timeout 900 python -m pytest tests/test_gpu_sharded.py -x -q -k "step_graph" > gpurun_out/r05_s9_pytest.txt 2>&1; grep -v amdgpu.ids gpurun_out/r05_s9_pytest.txt | grep -B2 -A25 "Error\|error" | head -120; tail -3 gpurun_out/r05_s9_pytest.txt
for f in 24 12 6 3; do
echo "== plain graph step, $f frames"
timeout 300 python bench.py --frames $f --steps 8 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('parallelism'))"
for g in 0 1; do
echo "== one rank on the sharded path, $f frames, --step-graph $g"
timeout 300 python bench.py --frames $f --steps 8 --warmup 3 --no-cpu-baseline --no-profile --single-rank-sharded --step-graph $g 2>gpurun_out/r05_s9_err_${f}_$g.txt | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('cfg_streams'), d['config'].get('step_graph'))" || tail -5 gpurun_out/r05_s9_err_${f}_$g.txt
done
done
} > $OUT 2>&1
cat $OUT
