#!/usr/bin/env python3
"""gpurun_out/pmc_summary.csv (tools/pmc_passes.sh + tools/pmc_reduce.py) -> the dominant kernel's HBM-side bytes per launch
as bench.py reports them in roofline.traffic:  python tools/pmc_traffic.py <pmc_summary.csv> <out.json> [commit]"""
import csv
import json
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    commit = sys.argv[3] if len(sys.argv) > 3 else "unknown"
    rows = [r for r in csv.DictReader(open(src)) if r["kernel"].startswith("hv_")]
    r = max(rows, key=lambda r: float(r["ms_profiled"]))
    n = int(r["launches"])
    fetch, write = float(r["fetch_GB"]) * 1e9 / n, float(r["write_GB"]) * 1e9 / n
    rec = {
        "kernel": r["kernel"], "launches_profiled": n, "fetch_bytes_per_launch": int(fetch), "write_bytes_per_launch": int(write),
        "bytes_per_launch": int(fetch + write), "mfma_util_pct_pmc": float(r["mfma_util_pct"] or 0),
        "eff_clock_GHz": float(r["eff_clock_GHz"] or 0), "commit": commit,
        "source": "tools/pmc_passes.sh on MI355X: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_VALU_MFMA_BUSY_CYCLES "
                  "GRBM_GUI_ACTIVE ... (three separate runs, each with --kernel-trace only) over `python bench.py --steps 2 --warmup 1 "
                  "--no-cpu-baseline --no-profile`; mean over every launch of the kernel in that run",
        "corrections": "FETCH_SIZE (KiB) x 1024 x 2 (gfx950: the counter reports half the bytes of wide coalesced reads; calibrated in "
                       "round 1 with tools/fillbw calib, 0.500-0.555 for this kernel's access pattern); WRITE_SIZE (KiB) x 1024 "
                       "(exact on tools/storebw)",
    }
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
