#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc ... --output-format csv runs to one per-kernel table.

usage: python tools/pmc_reduce.py <out.csv> <dir1> [<dir2> ...]
Each <dir> holds one counter pass (…_counter_collection.csv).  Per kernel name: launches, time under the
profiler, the summed counters, and derived figures:
  fetch_GB / write_GB        FETCH_SIZE, WRITE_SIZE are in KiB per dispatch; FETCH_SIZE x2 on gfx950 for wide
                             (16 B/lane) coalesced reads, as /opt/skills/guides/MI355X_MICROARCH.md prescribes
  hbm_GBps                   (fetch + write) / kernel time of that pass
  mfma_util_pct              SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 4 SIMDs * 256 CUs) * 100
                             (rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs: eff_clock_GHz = sum / 8 / kernel time)
"""
import collections
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:90]


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(dict)
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            seen = set()
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                c = r["Counter_Name"]
                agg[k][c] += float(r["Counter_Value"])
                key = (r["Dispatch_Id"], c)
                if key not in seen:
                    seen.add(key)
                    agg[k]["ns@" + c] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                    calls[k][c] = calls[k].get(c, 0) + 1
    counters = sorted({c for v in agg.values() for c in v if not c.startswith("ns@")})
    rows = []
    for k, v in agg.items():
        n = max(calls[k].values())
        ms = max(v.get("ns@" + c, 0.0) for c in counters) / 1e6
        rec = dict(kernel=k, launches=n, ms_profiled=round(ms, 3))
        for c in counters:
            rec[c] = v.get(c, 0.0)
        if "FETCH_SIZE" in v:
            rec["fetch_GB"] = round(2.0 * v["FETCH_SIZE"] * 1024 / 1e9, 4)
            rec["fetch_GBps"] = round(rec["fetch_GB"] / (v["ns@FETCH_SIZE"] / 1e9), 1) if v["ns@FETCH_SIZE"] else 0
        if "WRITE_SIZE" in v:
            rec["write_GB"] = round(v["WRITE_SIZE"] * 1024 / 1e9, 4)
            rec["write_GBps"] = round(rec["write_GB"] / (v["ns@WRITE_SIZE"] / 1e9), 1) if v["ns@WRITE_SIZE"] else 0
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("GRBM_GUI_ACTIVE"):
            rec["mfma_util_pct"] = round(100.0 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 4 * 256), 2)
            rec["eff_clock_GHz"] = round(v["GRBM_GUI_ACTIVE"] / 8 / v["ns@GRBM_GUI_ACTIVE"], 3) if v["ns@GRBM_GUI_ACTIVE"] else 0
        rows.append(rec)
    rows.sort(key=lambda r: -r["ms_profiled"])
    cols = ["kernel", "launches", "ms_profiled"] + counters + ["fetch_GB", "fetch_GBps", "write_GB", "write_GBps",
                                                               "mfma_util_pct", "eff_clock_GHz"]
    with open(out, "w") as fh:
        w = csv.DictWriter(fh, fieldnames=cols, extrasaction="ignore")
        w.writeheader()
        for r in rows[:40]:
            w.writerow(r)
    for r in rows[:14]:
        print({c: r[c] for c in cols if c in r and c not in counters})


if __name__ == "__main__":
    main()
