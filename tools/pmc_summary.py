#!/usr/bin/env python3
"""Sum PMC counters per kernel from a rocprofv3 rocpd database (one --pmc pass)."""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tables if t.startswith(p)][0]
pmc_ev, pmc_info, disp, sym = T("rocpd_pmc_event"), T("rocpd_info_pmc"), T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
cols = [r[1] for r in c.execute(f"pragma table_info({sym})")]
name_col = "display_name" if "display_name" in cols else "kernel_name"
q = (f"select s.{name_col}, i.name, e.value, d.id from {pmc_ev} e join {pmc_info} i on e.pmc_id = i.id "
     f"join {disp} d on e.event_id = d.event_id join {sym} s on d.kernel_id = s.id")
agg = defaultdict(lambda: defaultdict(float))
ndisp = defaultdict(set)
for kn, cn, val, did in c.execute(q):
    agg[kn][cn] += val
    ndisp[kn].add(did)
for kn, cs in agg.items():
    n = len(ndisp[kn])
    print(f"== {kn[:100]}  dispatches={n}")
    for cn, v in sorted(cs.items()):
        print(f"   {cn:32s} total={v:.4g}  per_dispatch={v / n:.4g}")
