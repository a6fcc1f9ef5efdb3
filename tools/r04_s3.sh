#!/bin/bash
# Round 4, GPU call 3 (~9 GPU-minutes): temporal kernel at three workgroups per CU, what the attention threshold test costs
# (timing probe build), and HBM-side fetch bytes of the convolution under the two workgroup rasters.
#   gpurun --timeout 1000 -- 'bash tools/r04_s3.sh'
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== kernel tests"; timeout 500 python -m pytest tests/test_gpu_kernels.py -q -x -k "temporal or attention" 2>&1 | tail -3
echo "== temporal"; timeout 200 python tools/microbench.py --only temporal 2>&1 | grep "^temporal"
echo "== attention: default library, then the probe build without the threshold test"
timeout 200 python tools/microbench.py --only attn 2>&1 | grep "^attention D=40"
HV_LIB=tools/bin/lib_attn_notest.so timeout 200 python tools/microbench.py --only attn 2>&1 | grep "^attention D=40"
REPO=$(pwd); cd /tmp
for m in 0 1; do
  rm -rf /tmp/pmc_conv_$m
  HUMANVID_TUNING=9=$m timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_conv_$m -- python $REPO/tools/microbench.py --only conv < /dev/null > $REPO/gpurun_out/r04_pmc_conv_$m.log 2>&1
  echo "== conv FETCH_SIZE, raster $m"; python $REPO/tools/pmc_fetch_by_kernel.py /tmp/pmc_conv_$m hv_conv3x3 | head -12
done
cd $REPO
for rep in 1 2; do for tune in "9=0" "9=2"; do HUMANVID_TUNING=$tune timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('step tuning=$tune', round(d['value'],3), round(d['ms_per_step'],2))"; done; done
} 2>&1 | tee gpurun_out/r04_s3.txt
