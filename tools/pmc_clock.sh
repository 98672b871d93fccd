cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pc; timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CU_CYCLES -d /tmp/pc -- python $R/tools/bench_attn.py --iters 3 --L 131072 > /tmp/pc.log 2>&1
python - <<PY
import sqlite3,glob
db=glob.glob("/tmp/pc/**/*.db",recursive=True)[0]
c=sqlite3.connect(db)
T=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
g=lambda p:[t for t in T if t.startswith(p)][0]
ev,info,disp,sym=g("rocpd_pmc_event"),g("rocpd_info_pmc"),g("rocpd_kernel_dispatch"),g("rocpd_info_kernel_symbol")
rows=c.execute(f"select s.display_name, i.name, e.value, d.start, d.end, d.id from {ev} e join {info} i on e.pmc_id=i.id join {disp} d on e.event_id=d.event_id join {sym} s on d.kernel_id=s.id").fetchall()
agg={}
for n,cn,v,st,en,did in rows:
    if "attn_partial" in n:
        a=agg.setdefault(did,{"dur":(en-st)/1e3}); a[cn]=a.get(cn,0)+v
for did,a in list(agg.items())[:4]:
    print(a, "MHz(GUI_ACTIVE/dur)=", a.get("GRBM_GUI_ACTIVE",0)/a["dur"])
PY
