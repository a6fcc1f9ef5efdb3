#!/bin/bash
# round 6, session 15: per-dispatch durations of hv_attention40 in bench.py --cfg-half 1 (which launches are the slow ones?)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
REPO=$(pwd)
export TMPDIR=/tmp
for args in "--cfg-half 1" ""; do
cd /tmp; rm -rf /tmp/prof_x
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_x -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile $args < /dev/null > /dev/null 2>&1
cd $REPO
DB=$(find /tmp/prof_x -name "*.db" | head -1)
python - "$DB" "$args" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select start, end-start from kernels where name like '%attention40%' order by start").fetchall()
print("bench.py", sys.argv[2], ": hv_attention40 dispatches in launch order, ms:", " ".join(f"{d/1e6:.2f}" for _, d in rows))
PY
done
