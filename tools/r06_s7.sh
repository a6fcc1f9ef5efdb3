#!/bin/bash
# round 6, session 7: per-shape step profiles (HV_PROFILE_DUMP) with the four-wave kernel on (default) / off (10=0)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for t in "" "10=0"; do
  tag=$( [ -z "$t" ] && echo w4 || echo w8 )
  HUMANVID_TUNING="$t" HV_PROFILE_DUMP=gpurun_out/r06_s7_step_profile_$tag.tsv timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
done
python - <<'PY'
def load(p):
    d={}
    for l in open(p).read().split('\n')[1:]:
        if not l.strip(): continue
        n,ms,key=l.split('\t')
        d[key]=(int(n),float(ms))
    return d
a=load('gpurun_out/r06_s7_step_profile_w4.tsv'); b=load('gpurun_out/r06_s7_step_profile_w8.tsv')
import re
def shape(k): return k.split('|')[1].strip() if '|' in k else k
sa={}; sb={}
for k,(n,ms) in a.items():
    if 'gemm' in k: sa[shape(k)]=(k.split('|')[0].strip(),n,ms)
for k,(n,ms) in b.items():
    if 'gemm' in k: sb[shape(k)]=(k.split('|')[0].strip(),n,ms)
tot=0
for s in sa:
    if s in sb and sa[s][0]!=sb[s][0]:
        print(f"{s:70s} {sb[s][0]:45s} {sb[s][2]:8.3f} -> {sa[s][0]:40s} {sa[s][2]:8.3f} ms  ({sa[s][1]} launches)")
        tot+=sa[s][2]-sb[s][2]
print('sum of differences (four-wave minus 8-wave):',tot)
print('total step kernels w4', sum(v[1] for v in a.values()), 'w8', sum(v[1] for v in b.values()))
PY
