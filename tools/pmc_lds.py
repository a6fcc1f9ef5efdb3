#!/usr/bin/env python3
"""LDS bank-conflict share per kernel from one rocprofv3 pass (`--pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace`):
cycles the LDS spent resolving bank conflicts / cycles it was active, summed over every dispatch of a kernel instantiation.

usage: python tools/pmc_lds.py <pass_dir> > profiles/rNN_lds_conflicts.txt
"""
import collections
import csv
import glob
import re
import sys

files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not files:
    raise SystemExit("no counter_collection.csv under " + sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(files[0])):
    name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE":
        n[name] += 1
print("# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per kernel instantiation over the bench command (tools/final_r06.sh, pass 4)")
print(f"{'dispatches':>10} {'conflict':>14} {'active':>14} {'ratio':>7}  kernel")
for name, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0.0)):
    a, b = c.get("SQ_LDS_BANK_CONFLICT", 0.0), c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    if b > 0:
        print(f"{n[name]:>10} {a:>14.0f} {b:>14.0f} {a / b:>7.3f}  {name}")
