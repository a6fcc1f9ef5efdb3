#!/bin/bash
# Round 5, session 1 (same box): the 8-interval GEMM loop (hv_gemm_p8_kernel, tuning key 8) against the two-group loop.
#   gpurun --timeout 1500 -- "HV_REF_SCRIPT_B64=$(base64 -w0 /root/reference/scripts/pose2vid.py) bash tools/r05_s1.sh"
mkdir -p gpurun_out
OUT=gpurun_out/r05_s1.txt
export TMPDIR=/tmp
{
echo "== bit-identity of the two loops at the bench shapes + the GEMM kernel tests (default = 8-interval loop)"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" 2>&1 | tail -4
for glds in 1 2; do for p8 in 0 1; do
  echo "== microbench p8=$p8 selection=$glds"
  HV_GEMM_P8=$p8 HV_GEMM_GLDS=$glds timeout 300 python tools/microbench.py --only gemm 2>&1 | grep "^gemm "
done; done
echo "== microbench p8=1 selection=1, no s_setprio around the MFMA groups"
HV_LIB=tools/bin/lib_noprio.so HV_GEMM_P8=1 timeout 300 python tools/microbench.py --only gemm 2>&1 | grep "^gemm "
echo "== step A/B (graph replay, 10 steps)"
for i in 1 2; do for t in "8=0" "8=1"; do
  echo -n "step tuning=$t "
  HUMANVID_TUNING=$t timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'])"
done; done
echo "== full-size forward parity under the default (8-interval) loop"
timeout 600 python -m pytest tests/test_gpu_fullsize_parity.py -x -q -s -k "config3" 2>&1 | grep -i "nrmse\|passed\|failed" | head -5
} > $OUT 2>&1
{
echo "# tests/test_gpu_script_contract.py on MI355X (round 5 tree): /root/reference/scripts/pose2vid.py executed UNMODIFIED with runpy"
echo "# (script text handed over in HV_REF_SCRIPT_B64 on the gpurun command line; not part of this repository)"
timeout 600 python -m pytest tests/test_gpu_script_contract.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -8
} > gpurun_out/r05_script_contract.txt 2>&1
HV_PROFILE_DUMP=gpurun_out/r05_s1_step_profile.tsv timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r05_s1_bench.json 2>/dev/null
tail -60 $OUT
