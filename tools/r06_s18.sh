#!/bin/bash
# round 6, session 18: the four-wave kernel's deferred residual form (level-2 ff2) + level-2 QKV: hardware bit-identity, then the
# per-shape step profiles with the kernel on (default) / off (10=0)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "== kernel tests"
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "four_wave or gemm_epilogue_forms or bench_shape_gemm" 2>&1 | tail -3
bash tools/r06_s7.sh 2>&1 | grep "^{" | cut -c1-160
python - <<'PY'
def load(p):
    d={}
    for l in open(p).read().split('\n')[1:]:
        if not l.strip() or l.startswith('#'): continue
        parts=l.split('\t')
        try: n=int(parts[0]); ms=float(parts[1])
        except: continue
        d[parts[2] if len(parts)>2 else '']=(n,ms)
    return d
a=load('gpurun_out/r06_s7_step_profile_w4.tsv'); b=load('gpurun_out/r06_s7_step_profile_w8.tsv')
def shape(k): return k.split('|')[1].strip() if '|' in k else k
sa={}; sb={}
for k,(n,ms) in a.items():
    if 'gemm' in k: sa[shape(k)]=(k.split('|')[0].strip(),n,ms)
for k,(n,ms) in b.items():
    if 'gemm' in k: sb[shape(k)]=(k.split('|')[0].strip(),n,ms)
tot=0
for s in sa:
    if s in sb and sa[s][0]!=sb[s][0]:
        print(f"{s:62s} {sb[s][0][:38]:38s} {sb[s][2]:7.3f} -> {sa[s][0][:34]:34s} {sa[s][2]:7.3f} ms ({sa[s][1]})")
        tot+=sa[s][2]-sb[s][2]
print('sum of differences (four-wave minus 8-wave):',tot)
print('total ms w4', sum(v[1] for v in a.values()), 'w8', sum(v[1] for v in b.values()))
PY
