#!/bin/bash
# Counter passes over the bench command (one pass per counter group; --pmc only with --kernel-trace, as the
# pool requires).  Run on the GPU box from the repo root; writes gpurun_out/pmc_{fetch,write,mfma}/ and the
# reduced gpurun_out/pmc_summary.csv.
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_fetch -- $CMD < /dev/null > $REPO/gpurun_out/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_write -- $CMD < /dev/null > $REPO/gpurun_out/pmc_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_mfma -- $CMD < /dev/null > $REPO/gpurun_out/pmc_mfma.log 2>&1
cd $REPO
python tools/pmc_reduce.py gpurun_out/pmc_summary.csv gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma
# keep only the reduced table (the raw per-dispatch CSVs are tens of MB)
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma
tail -3 gpurun_out/pmc_*.log
