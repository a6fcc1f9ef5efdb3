#!/bin/bash
# counters of single GEMM shapes (tile policies 9 / 6) -- one rocprofv3 --pmc pass per counter group, kernel-trace only
mkdir -p gpurun_out/pmcg
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o -E "\b(TCC_HIT_sum|TCC_MISS_sum|TCC_REQ_sum|TCC_EA0_RDREQ_sum|TCC_EA0_RDREQ_32B_sum|TA_BUSY_avr|TA_BUSY_max|TCP_PENDING_STALL_CYCLES_sum|TCP_TCC_READ_REQ_sum|TCP_TOTAL_CACHE_ACCESSES_sum|TCP_TCC_READ_REQ_LATENCY_sum|TCP_GATE_EN1_sum|TCP_GATE_EN2_sum|TA_ADDR_STALLED_BY_TC_CYCLES_sum|TA_DATA_STALLED_BY_TC_CYCLES_sum|TCP_TA_TCP_STATE_READ_sum|SQ_WAIT_INST_ANY|SQ_INSTS_VMEM_RD|SQ_ACTIVE_INST_VMEM|SQ_INST_LEVEL_VMEM|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_INSTS_LDS|SQ_ACTIVE_INST_LDS|SQ_LDS_BANK_CONFLICT|SQ_LDS_IDX_ACTIVE|GRBM_GUI_ACTIVE|TCC_TAG_STALL_sum|TCC_BUSY_sum|TCP_TCC_NC_READ_REQ_sum|TCC_EA0_RD_UNCACHED_32B_sum)\b" | sort -u > $REPO/gpurun_out/pmcg/avail.txt
cat $REPO/gpurun_out/pmcg/avail.txt | tr '\n' ' '; echo
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for pol in 9 6; do
    HV_GEMM_GLDS=$pol HV_MB_ONLY_L0=1 timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $REPO/gpurun_out/pmcg/g${i}_p$pol -- python $REPO/tools/microbench.py --only gemm --images 48 < /dev/null > $REPO/gpurun_out/pmcg/g${i}_p$pol.log 2>&1
  done
done
cd $REPO
python - <<'PY'
import csv,glob,collections,re,os
for d in sorted(glob.glob('gpurun_out/pmcg/g*_p*')):
    if not os.path.isdir(d): continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob(d+'/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            k=re.sub(r'\(.*$','',r['Kernel_Name'])[:60]
            if 'hv_gemm' not in k: continue
            key=(k, r.get('Grid_Size') or r.get('Workgroup_Size'))
            agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
    for k in agg:
        print(d[-6:], k, {c: '%.3g'%(v/cnt[(k,c)]) for c,v in agg[k].items()}, 'n=%d'%max(cnt[(k,c)] for c in agg[k]))
PY
rm -rf gpurun_out/pmcg/g*_p*/  # raw csv are large
