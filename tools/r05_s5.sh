#!/bin/bash
# Round 5, session 5: attention40 with two tiles per barrier + conflict-free constant rows (new) against the round-4 kernel (old),
# same box; kernel tests of the new one; single-rank-sharded step at the R = 2 / 4 / 8 per-rank frame counts.
mkdir -p gpurun_out
OUT=gpurun_out/r05_s5.txt
{
echo "== attention kernel tests (new kernel)"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" 2>&1 | tail -3
for i in 1 2; do
echo "== microbench attn: old (round-4 kernel)"
HV_LIB=tools/bin/lib_attn_old.so timeout 200 python tools/microbench.py --only attn 2>&1 | grep -i "attn\|attention" | head -8
echo "== microbench attn: new"
timeout 200 python tools/microbench.py --only attn 2>&1 | grep -i "attn\|attention" | head -8
done
echo "== step A/B"
for i in 1 2; do for lib in tools/bin/lib_attn_old.so humanvid_amd/lib/libhumanvid_hip.so; do
  echo -n "step lib=$lib "
  HUMANVID_HIP_LIB=$lib timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'])"
done; done
echo "== single-rank sharded step at the per-rank frame counts of R = 1 / 2 / 4 / 8 (24 / 12 / 6 / 3 frames)"
for f in 24 12 6 3; do
  echo -n "frames=$f plain-graph: "
  timeout 300 python bench.py --frames $f --steps 8 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'])"
  echo -n "frames=$f single-rank-sharded (serial halves): "
  timeout 300 python bench.py --frames $f --single-rank-sharded --steps 8 --warmup 3 --no-cpu-baseline --no-profile 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['config'].get('cfg_streams'))"
done
} > $OUT 2>&1
cat $OUT
