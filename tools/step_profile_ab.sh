#!/bin/bash
# Same-box A/B of the whole step BY GEMM SHAPE: two eagerly launched steps with HIP events around every launch
# (HV_PROFILE_DUMP), one per tuning string, then the shapes whose kernel selection differs with their summed launch time.
#   bash tools/step_profile_ab.sh <tag> "<HUMANVID_TUNING a>" "<HUMANVID_TUNING b>"     e.g.  ... r06_s18 "" "10=0"
# -> gpurun_out/<tag>_step_profile_a.tsv, _b.tsv and the table on stdout (profiles/r06_s18.txt was made this way)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tag=$1; ta=$2; tb=$3
HUMANVID_TUNING="$ta" HV_PROFILE_DUMP=gpurun_out/${tag}_step_profile_a.tsv timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | cut -c1-160
HUMANVID_TUNING="$tb" HV_PROFILE_DUMP=gpurun_out/${tag}_step_profile_b.tsv timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | cut -c1-160
python - "$tag" <<'PY'
import sys
def load(p):
    d = {}
    for l in open(p).read().split('\n')[1:]:
        if not l.strip() or l.startswith('#'):
            continue
        parts = l.split('\t')
        try:
            n, ms = int(parts[0]), float(parts[1])
        except ValueError:
            continue
        key = parts[2] if len(parts) > 2 else ''
        if 'gemm' in key and '|' in key:
            d[key.split('|')[1].strip()] = (key.split('|')[0].strip(), n, ms)
    return d
tag = sys.argv[1]
a, b = load(f'gpurun_out/{tag}_step_profile_a.tsv'), load(f'gpurun_out/{tag}_step_profile_b.tsv')
tot = 0.0
for s in a:
    if s in b and a[s][0] != b[s][0]:
        print(f"{s:62s} b: {b[s][0][:38]:38s} {b[s][2]:7.3f} -> a: {a[s][0][:34]:34s} {a[s][2]:7.3f} ms ({a[s][1]} launches)")
        tot += a[s][2] - b[s][2]
print(f"sum of differences (a minus b): {tot:+.3f} ms per step")
PY
