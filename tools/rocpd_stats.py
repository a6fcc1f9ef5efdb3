#!/usr/bin/env python3
"""Reduce a rocprofv3 rocpd database (ROCm 7.2 default output) to a per-kernel stats CSV.
usage: python tools/rocpd_stats.py <results.db> <out.csv> ["header comment"]"""
import re
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, "
                       "max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = [f"# {note}", f"# total kernel time {tot:.1f} ms", "name,calls,total_ms,avg_us,min_us,max_us,pct"]
    for r in rows[:60]:
        n = re.sub(r"\(.*$", "", r[0])[:100].replace(",", ";")
        lines.append(f"{n},{r[1]},{r[2]:.2f},{r[3]:.1f},{r[4]:.1f},{r[5]:.1f},{100 * r[2] / tot:.1f}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:25]))


if __name__ == "__main__":
    main()
