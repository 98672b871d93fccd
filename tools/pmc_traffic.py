#!/usr/bin/env python3
"""HBM traffic per launch of the kernels whose name contains SUBSTR, from two rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE, separate runs of the same command).  gfx950 correction per
MI355X_MICROARCH.md section HBM: FETCH_SIZE counts 128-B requests of wide coalesced streams as 64 B -> doubled;
WRITE_SIZE is uncalibrated (reported as is).

    python tools/pmc_traffic.py FETCH.db WRITE.db SUBSTR [out.json]
"""
import json
import sqlite3
import sys


def per_kernel(db, counter, substr):
    c = sqlite3.connect(db)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tables if t.startswith(p)][0]
    pmc_ev, pmc_info, disp, sym = T("rocpd_pmc_event"), T("rocpd_info_pmc"), T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
    cols = [r[1] for r in c.execute(f"pragma table_info({sym})")]
    name_col = "display_name" if "display_name" in cols else "kernel_name"
    q = (f"select s.{name_col}, e.value, d.id from {pmc_ev} e join {pmc_info} i on e.pmc_id = i.id "
         f"join {disp} d on e.event_id = d.event_id join {sym} s on d.kernel_id = s.id where i.name = ?")
    tot, ids = 0.0, set()
    for kn, val, did in c.execute(q, (counter,)):
        if substr in kn:
            tot += val
            ids.add(did)
    return tot, len(ids)


def main():
    fdb, wdb, substr = sys.argv[1:4]
    f_kb, nf = per_kernel(fdb, "FETCH_SIZE", substr)
    w_kb, nw = per_kernel(wdb, "WRITE_SIZE", substr)
    out = {
        "kernel": substr, "launches_fetch_pass": nf, "launches_write_pass": nw,
        "FETCH_SIZE_KB_per_launch": round(f_kb / max(nf, 1), 1), "WRITE_SIZE_KB_per_launch": round(w_kb / max(nw, 1), 1),
        "correction": "FETCH_SIZE doubled (gfx950 rocprofv3 tallies the 128-B requests of wide coalesced streams at 64 B, "
                      "MI355X_MICROARCH.md section HBM); WRITE_SIZE uncorrected (uncalibrated)",
        "hbm_read_bytes_per_launch": round(2 * f_kb * 1024 / max(nf, 1)),
        "hbm_write_bytes_per_launch": round(w_kb * 1024 / max(nw, 1)),
    }
    out["hbm_bytes_per_launch"] = out["hbm_read_bytes_per_launch"] + out["hbm_write_bytes_per_launch"]
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 4:
        json.dump(out, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
