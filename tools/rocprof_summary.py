#!/usr/bin/env python3
"""Per-kernel duration summary from a rocprofv3 rocpd SQLite database (the default output
of `rocprofv3 --kernel-trace --stats`), or from a kernel-trace CSV.  Prints a CSV table:
kernel, calls, total_us, avg_us, min_us, max_us, pct."""
import csv
import sqlite3
import sys
from collections import defaultdict


def from_db(path):
    c = sqlite3.connect(path)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({sym})")]
    name_col = "display_name" if "display_name" in cols else ("kernel_name" if "kernel_name" in cols else "name")
    rows = c.execute(f"select s.{name_col}, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id").fetchall()
    return [(n, (e - s) / 1e3) for n, s, e in rows]


def from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    return out


def main():
    path = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = from_db(path) if path.endswith(".db") else from_csv(path)
    agg = defaultdict(list)
    for n, us in rows:
        agg[n].append(us)
    tot = sum(sum(v[skip:]) for v in agg.values()) or 1.0
    print("kernel,calls,total_us,avg_us,min_us,max_us,pct")
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        v = v[skip:] if len(v) > skip else v
        short = n if len(n) < 110 else n[:107] + "..."
        print(f"\"{short}\",{len(v)},{sum(v):.1f},{sum(v)/len(v):.2f},{min(v):.2f},{max(v):.2f},{100*sum(v)/tot:.1f}")


if __name__ == "__main__":
    main()
