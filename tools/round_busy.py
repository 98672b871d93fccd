#!/usr/bin/env python3
"""GPU busy fraction of the tree rounds from a rocprofv3 kernel-trace database: over the span between the 3rd and
the last tree_commit_kernel, sum of kernel durations / wall span, and the distribution of the gaps between
consecutive kernels.   python tools/round_busy.py <trace.db>"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({sym})")]
    name_col = "display_name" if "display_name" in cols else ("kernel_name" if "kernel_name" in cols else "name")
    rows = c.execute(f"select s.{name_col}, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id order by d.start").fetchall()
    commits = [i for i, r in enumerate(rows) if "tree_commit_kernel" in r[0]]
    if len(commits) < 5:
        print("not enough rounds")
        return
    lo, hi = commits[2], commits[-1]
    span = rows[lo + 1: hi + 1]
    rounds = len(commits) - 3
    t0, t1 = rows[lo][2], rows[hi][2]
    busy = sum(e - s for _, s, e in span)
    gaps = [max(0, span[i + 1][1] - span[i][2]) for i in range(len(span) - 1)]
    gaps.sort()
    print(f"rounds {rounds}  kernels/round {len(span) / rounds:.1f}  wall {(t1 - t0) / rounds / 1e6:.3f} ms/round  "
          f"busy {busy / rounds / 1e6:.3f} ms/round ({100 * busy / (t1 - t0):.1f} %)")
    n = len(gaps)
    print(f"gaps us: median {gaps[n // 2] / 1e3:.2f}  p90 {gaps[int(n * .9)] / 1e3:.2f}  p99 {gaps[int(n * .99)] / 1e3:.2f}  "
          f"max {gaps[-1] / 1e3:.1f}  sum/round {sum(gaps) / rounds / 1e6:.3f} ms")
    big = sum(g for g in gaps if g > 20e3)
    print(f"gaps > 20 us: {sum(1 for g in gaps if g > 20e3) / rounds:.1f}/round, {big / rounds / 1e6:.3f} ms/round (the round's host read)")


if __name__ == "__main__":
    main()
