#!/bin/bash
# Round 5, session 6: storage-model parity test; attention40 conflict-free constant rows alone; step profiles at 3 / 6 frames.
mkdir -p gpurun_out
OUT=gpurun_out/r05_s6.txt
{
echo "== native forward vs the storage-model golden (config 3)"
timeout 900 python -m pytest tests/test_gpu_storage_model.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -45
for i in 1 2; do
echo "== microbench attn: old (round-4 kernel)"
HV_LIB=tools/bin/lib_attn_old.so timeout 200 python tools/microbench.py --only attn 2>&1 | grep "attention D=40" | head -3
echo "== microbench attn: conflict-free constant rows, one tile per barrier"
timeout 200 python tools/microbench.py --only attn 2>&1 | grep "attention D=40" | head -3
done
for f in 3 6; do
HV_PROFILE_DUMP=gpurun_out/r05_s6_step_profile_f$f.tsv timeout 300 python bench.py --frames $f --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r05_s6_bench_f$f.json 2>/dev/null
done
} > $OUT 2>&1
cat $OUT
