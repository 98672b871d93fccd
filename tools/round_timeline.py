#!/usr/bin/env python3
"""Per-pass launch timeline of the tree rounds from a rocprofv3 kernel-trace database (where does a draft
pass spend its time, launch by launch?).  A round = the kernels between two tree_commit_kernel launches; its passes are cut at
tree_grow_kernel (ends draft pass 0..4) and tree_verify_inputs_kernel (starts the verification pass).  For every position of
the round's launch sequence: kernel, mean duration, mean gap to the previous kernel's end -- averaged over the rounds of the
trace that have the modal launch count -- and per pass: launches, sum of durations, sum of gaps, wall.
    python tools/round_timeline.py <trace.db> [out.json]"""
import collections
import json
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z_0-9:]+)(<[^(]*>)?", n)
    base = m.group(1) if m else n
    tpl = (m.group(2) or "") if m else ""
    tpl = tpl.replace("ElemF16", "f16").replace("ElemBF16", "bf16").replace(" ", "")
    return (base.split("::")[-1] + tpl)[:60]


def main():
    c = sqlite3.connect(sys.argv[1])
    tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({sym})")]
    name_col = "display_name" if "display_name" in cols else ("kernel_name" if "kernel_name" in cols else "name")
    rows = c.execute(f"select s.{name_col}, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id order by d.start").fetchall()
    commits = [i for i, r in enumerate(rows) if "tree_commit_kernel" in r[0]]
    rounds = [rows[commits[i] + 1: commits[i + 1] + 1] for i in range(len(commits) - 1)]
    if not rounds:
        print("no rounds in the trace")
        return
    modal = collections.Counter(len(r) for r in rounds).most_common(1)[0][0]
    # only replayed / steady rounds: modal launch count and no gap > 50 us inside (the eager, event-bracketed ones have more launches)
    good = [r for r in rounds if len(r) == modal and max(r[i + 1][1] - r[i][2] for i in range(len(r) - 1)) < 50e3]
    if not good:
        good = [r for r in rounds if len(r) == modal]
    n = len(good)
    seq = []
    for pos in range(modal):
        name = short(good[0][pos][0])
        dur = sum(r[pos][2] - r[pos][1] for r in good) / n / 1e3
        gap = sum((r[pos][1] - r[pos - 1][2]) if pos else 0 for r in good) / n / 1e3
        seq.append({"kernel": name, "us": round(dur, 2), "gap_before_us": round(gap, 2)})
    passes, cur, label = [], [], 0
    for e in seq:
        if e["kernel"].startswith("tree_verify_inputs_kernel"):
            if cur:
                passes.append(("between", cur))
            cur = [e]
            continue
        cur.append(e)
        if e["kernel"].startswith("tree_grow_kernel"):
            passes.append((f"draft pass {label}", cur))
            cur, label = [], label + 1
    if cur:
        passes.append(("verification pass + collapse + commit", cur))
    out = {"rounds_averaged": n, "launches_per_round": modal,
           "round_wall_us": round(sum(r[-1][2] - r[0][1] for r in good) / n / 1e3, 1), "passes": []}
    for name, ks in passes:
        by = collections.OrderedDict()
        for e in ks:
            b = by.setdefault(e["kernel"], {"launches": 0, "us": 0.0})
            b["launches"] += 1
            b["us"] = round(b["us"] + e["us"], 2)
        out["passes"].append({"pass": name, "launches": len(ks), "kernel_us": round(sum(e["us"] for e in ks), 1),
                              "gap_us": round(sum(e["gap_before_us"] for e in ks), 1),
                              "wall_us": round(sum(e["us"] + e["gap_before_us"] for e in ks), 1),
                              "by_kernel": by, "sequence": ks if name.startswith("draft pass") and name[-1] in "01" else None})
    js = json.dumps(out, indent=1)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(js)
    for p in out["passes"]:
        print(f"{p['pass']:42s} launches {p['launches']:4d}  kernels {p['kernel_us']:8.1f} us  gaps {p['gap_us']:6.1f} us  wall {p['wall_us']:8.1f} us")


if __name__ == "__main__":
    main()
