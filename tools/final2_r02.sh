#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py > gpurun_out/final2_bench.txt 2> gpurun_out/final2_bench.err; tail -c 400 gpurun_out/final2_bench.txt | head -c 200; echo
HV_PROFILE_DUMP=gpurun_out/final2_step_profile.tsv timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/final2_bench2.txt 2>&1
bash tools/prof_bench.sh final2_r02 --no-profile > gpurun_out/final2_prof_head.txt 2>&1; head -12 gpurun_out/final2_prof_head.txt
timeout 600 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-profile --fp8-attention 0 2>&1 | tail -1 | cut -c1-200
timeout 600 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>&1 | tail -1 | cut -c1-200
timeout 300 python bench.py --config 2 --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | tail -1 | cut -c1-200
