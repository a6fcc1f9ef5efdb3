#!/bin/bash
# Round 5, session 15: the -m gpu suite on the final tree after trimming the storage-model test (frame subsets for the two
# largest transformer blocks), with per-test durations.
mkdir -p gpurun_out
timeout 2000 python -m pytest tests -m gpu -q -x --durations=30 > gpurun_out/r05_final3_pytest.txt 2>&1
grep -v "amdgpu.ids" gpurun_out/r05_final3_pytest.txt | tail -40
