mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" 2>&1 | tail -3
for v in tools/bin/lib_gemm_base.so humanvid_amd/lib/libhumanvid_hip.so; do HV_LIB=$v timeout 300 python tools/microbench.py --only gemm 2>&1 | grep -i "geglu" | awk -v v=$(basename $v) '{printf "%-22s %s\n", v, $0}'; done
for rep in 1 2; do for v in tools/bin/lib_gemm_base.so humanvid_amd/lib/libhumanvid_hip.so; do HUMANVID_HIP_LIB=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('step $v', d['value'], d['ms_per_step'])"; done; done
} | tee gpurun_out/r03_gelu_ab.txt
