#!/usr/bin/env python3
"""Counter passes -> HBM-side traffic PER KERNEL AND PER SHAPE of one denoising step.

usage: python tools/pmc_by_shape.py <step_profile.tsv> <out.json> <fetch_dir> <write_dir> <mfma_dir>

<step_profile.tsv>  bench.py's HV_PROFILE_DUMP of the same build: one line per "kernel variant | shape" key plus the "#seq"
                    line = the launch order of one step as indices into those lines (hv_profile_end).
<*_dir>             one rocprofv3 pass each (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, `--pmc SQ_VALU_MFMA_BUSY_CYCLES
                    GRBM_GUI_ACTIVE`, every one with --kernel-trace only) over `python bench.py --steps 2 --warmup 1
                    --no-cpu-baseline --no-profile`.
The dispatches of the LAST complete step of a pass (it ends with hv_cfg_ddim_kernel + hv_clear_kernel) are lined up with the
launch order one to one -- the step is a fixed launch sequence, eager, graph-replayed or profiled -- which attaches the shape
to every dispatch; kernel names are cross-checked.  Per key: launches, FETCH_SIZE x 2 (gfx950: the counter reports half the
bytes of wide coalesced reads, MI355X_MICROARCH.md section HBM) and WRITE_SIZE per launch, against the algorithmic read /
written bytes (bench.price_launch_rw), and the MFMA-busy share of the kernel's own active cycles.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def base(name):
    name = re.sub(r"^void ", "", name)
    return re.split(r"[<( ]", name, 1)[0]


def last_step(dir_, counters, nseq):
    files = glob.glob(dir_ + "/**/*counter_collection.csv", recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {dir_}")
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(files[0])):
        d = disp.setdefault(int(r["Dispatch_Id"]), dict(name=r["Kernel_Name"], ns=float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    order = [disp[k] for k in sorted(disp)]
    ends = [i for i, d in enumerate(order) if base(d["name"]) == "hv_clear_kernel"]
    if len(ends) < 2:
        raise SystemExit("fewer than two step ends (hv_clear_kernel) in " + files[0])
    step = order[ends[-1] + 1 - nseq:ends[-1] + 1]
    if len(step) != nseq:
        raise SystemExit(f"{files[0]}: {len(step)} dispatches for a step of {nseq} launches")
    return step


def main():
    prof, out, d_fetch, d_write, d_mfma = sys.argv[1:6]
    import bench  # price_launch_rw, csrc_digest

    keys, seq = [], None
    for line in open(prof).read().splitlines()[1:]:
        if line.startswith("#seq"):
            seq = [int(x) for x in line.split("\t", 1)[1].split(",")]
        elif line and not line.startswith("#"):
            keys.append(line.split("\t", 2)[2])
    if seq is None:
        raise SystemExit("no #seq line in " + prof)
    passes = {"FETCH_SIZE": last_step(d_fetch, ["FETCH_SIZE"], len(seq)), "WRITE_SIZE": last_step(d_write, ["WRITE_SIZE"], len(seq)),
              "MFMA": last_step(d_mfma, ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"], len(seq))}
    agg = collections.OrderedDict()
    for pos, ki in enumerate(seq):
        key = keys[ki]
        kb = base(key.partition(" | ")[0])
        for pname, step in passes.items():
            # (a note names its launcher's kernel family: hv_ln_stats_kernel files hv_ln_stats_rows_kernel<8> launches)
            if base(step[pos]["name"])[:9] != kb[:9] and kb != "other":
                raise SystemExit(f"launch {pos}: profile says {kb}, {pname} pass says {step[pos]['name'][:60]}")
        a = agg.setdefault(key, collections.defaultdict(float))
        a["launches"] += 1
        a["fetch"] += 2.0 * 1024.0 * passes["FETCH_SIZE"][pos].get("FETCH_SIZE", 0.0)
        a["write"] += 1024.0 * passes["WRITE_SIZE"][pos].get("WRITE_SIZE", 0.0)
        a["mfma_busy"] += passes["MFMA"][pos].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        a["gui_active"] += passes["MFMA"][pos].get("GRBM_GUI_ACTIVE", 0.0)
        a["ns"] += passes["MFMA"][pos]["ns"]
    shapes, kernels = [], collections.OrderedDict()
    for key, a in agg.items():
        n = a["launches"]
        rd, wr = bench.price_launch_rw(key)
        rec = dict(key=key, launches=int(n), fetch_bytes_per_launch=int(a["fetch"] / n), write_bytes_per_launch=int(a["write"] / n),
                   algorithmic_read_bytes=int(rd), algorithmic_write_bytes=int(wr),
                   fetch_over_algorithmic=round(a["fetch"] / n / rd, 3) if rd else None,
                   write_over_algorithmic=round(a["write"] / n / wr, 3) if wr else None,
                   mfma_busy_pct=round(100.0 * a["mfma_busy"] / (a["gui_active"] / 8 * 4 * 256), 2) if a["gui_active"] else None,
                   avg_us_profiled=round(a["ns"] / n / 1e3, 1))
        shapes.append(rec)
        k = kernels.setdefault(key.partition(" | ")[0], collections.defaultdict(float))
        for f in ("launches", "fetch", "write", "mfma_busy", "gui_active", "ns"):
            k[f] += a[f]
        k["alg_r"] += rd * n
        k["alg_w"] += wr * n
    kout = collections.OrderedDict()
    for name, k in sorted(kernels.items(), key=lambda kv: -kv[1]["ns"]):
        n = k["launches"]
        kout[name] = dict(launches_per_step=int(n), fetch_bytes_per_launch=int(k["fetch"] / n), write_bytes_per_launch=int(k["write"] / n),
                          algorithmic_read_bytes_per_launch=int(k["alg_r"] / n), algorithmic_write_bytes_per_launch=int(k["alg_w"] / n),
                          mfma_busy_pct=round(100.0 * k["mfma_busy"] / (k["gui_active"] / 8 * 4 * 256), 2) if k["gui_active"] else None,
                          eff_clock_GHz=round(k["gui_active"] / 8 / k["ns"], 3) if k["ns"] else None,
                          ms_per_step_profiled=round(k["ns"] / 1e6, 3))
    shapes.sort(key=lambda r: -(r["avg_us_profiled"] * r["launches"]))
    rec = dict(csrc_digest=bench.csrc_digest(), kernels=kout, shapes=shapes[:60],
               source="tools/final_r06.sh on MI355X: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_VALU_MFMA_BUSY_CYCLES "
                      "GRBM_GUI_ACTIVE (three separate runs, each with --kernel-trace only) over `python bench.py --steps 2 --warmup 1 "
                      "--no-cpu-baseline --no-profile`; the dispatches of the last complete step lined up with the step's launch order "
                      "(hv_profile_end '#seq')",
               corrections="FETCH_SIZE (KiB) x 1024 x 2 (gfx950: the counter reports half the bytes of wide coalesced reads: "
                           "MI355X_MICROARCH.md, calibrated in round 1 with tools/fillbw: 0.500-0.555); WRITE_SIZE (KiB) x 1024 (exact "
                           "on tools/storebw); Infinity-Cache hits are counted as traffic")
    json.dump(rec, open(out, "w"), indent=1)
    for name, k in list(kout.items())[:8]:
        print(name, k)
    for r in shapes[:12]:
        print(r)


if __name__ == "__main__":
    main()
