#!/bin/bash
mkdir -p gpurun_out/pmcg
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for pf in 0 2; do
  HV_GEMM_PF=$pf HV_MB_ONLY_L0=1 timeout 120 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum --kernel-trace --output-format csv -d $REPO/gpurun_out/pmcg/pf$pf -- python $REPO/tools/microbench.py --only gemm < /dev/null > $REPO/gpurun_out/pmcg/pf$pf.log 2>&1
  HV_GEMM_PF=$pf HV_MB_ONLY_L0=1 timeout 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $REPO/gpurun_out/pmcg/pfb$pf -- python $REPO/tools/microbench.py --only gemm < /dev/null > $REPO/gpurun_out/pmcg/pfb$pf.log 2>&1
done
cd $REPO
python - <<'PY'
import csv,glob,collections,re,os
for d in sorted(glob.glob('gpurun_out/pmcg/pf*')):
    if not os.path.isdir(d): continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); tm=collections.defaultdict(float)
    for f in glob.glob(d+'/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            k=re.sub(r'\(.*$','',r['Kernel_Name'])[:60]
            if 'hv_gemm' not in k: continue
            agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
    for k in agg:
        print(d[-5:], k, {c: '%.3g'%(v/cnt[(k,c)]) for c,v in agg[k].items()}, 'n=%d'%max(cnt[(k,c)] for c in agg[k]))
PY
rm -rf gpurun_out/pmcg/pf*/
