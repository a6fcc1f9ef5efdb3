#!/usr/bin/env python3
"""FETCH_SIZE per kernel instantiation and launch from one rocprofv3 pass (`--pmc FETCH_SIZE --kernel-trace`): KiB x 1024, x 2 for
gfx950's half-counted wide reads (MI355X_MICROARCH.md, section HBM; the corrections of tools/pmc_by_shape.py): python tools/pmc_fetch_by_kernel.py <pass_dir> [name filter]"""
import collections
import csv
import glob
import re
import sys

files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not files:
    raise SystemExit("no counter_collection.csv under " + sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc, n = collections.defaultdict(float), collections.Counter()
for r in csv.DictReader(open(files[0])):
    if r["Counter_Name"] != "FETCH_SIZE":
        continue
    name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
    grid = r.get("Grid_Size", "")
    key = f"{name} grid={grid}"
    acc[key] += float(r["Counter_Value"])
    n[key] += 1
for key, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    if flt in key:
        print(f"{n[key]:>5} launches  {2.0 * 1024.0 * v / n[key] / 1e6:>10.1f} MB fetched per launch  {key}")
