# PMC profile of the two roofline kernels INSIDE bench.py's rounds (separate passes per counter group; kernel trace only).
# Writes gpurun_out/pmc_round_kernels.json: counters per launch + derived utilisations.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="${BENCH_ARGS:---steps 4 --warmup 1 --no-vanilla --no-cpu-baseline --no-graphs}"
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES"; do
  i=$((i+1)); rm -rf /tmp/prk_$i
  timeout 400 rocprofv3 --kernel-trace --pmc $grp -d /tmp/prk_$i -- python $R/bench.py $ARGS > /tmp/prk_$i.log 2>&1
done
python - <<PY
import glob, json, sqlite3
from collections import defaultdict
KERNELS = {"attention_ws": "attn_partial_ws_kernel", "gemm_gate_up_M74": "skinny_gemm_kernel<ElemF16, 5, 8, 1,",
           "gemm_qkv_rope_M74": "skinny_gemm_kernel<ElemF16, 5, 4, 2,", "gemm_lm_head_draft": None}
out = {k: {} for k in KERNELS if KERNELS[k]}
for db in sorted(glob.glob("/tmp/prk_*/**/*.db", recursive=True)):
    c = sqlite3.connect(db)
    T = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    g = lambda p: [t for t in T if t.startswith(p)][0]
    ev, info, disp, sym = g("rocpd_pmc_event"), g("rocpd_info_pmc"), g("rocpd_kernel_dispatch"), g("rocpd_info_kernel_symbol")
    cols = [r[1] for r in c.execute(f"pragma table_info({sym})")]
    nc = "display_name" if "display_name" in cols else "kernel_name"
    rows = c.execute(f"select s.{nc}, i.name, e.value, d.id, d.start, d.end from {ev} e join {info} i on e.pmc_id=i.id "
                     f"join {disp} d on e.event_id=d.event_id join {sym} s on d.kernel_id=s.id").fetchall()
    for key, sub in KERNELS.items():
        if not sub:
            continue
        agg, ids, dur = defaultdict(float), set(), {}
        for n, cn, v, did, st, en in rows:
            if sub in n:
                agg[cn] += v; ids.add(did); dur[did] = (en - st) / 1e3
        if ids:
            for cn, v in agg.items():
                out[key][cn] = round(v / len(ids), 1)
            out[key].setdefault("launches", len(ids))
            out[key]["avg_us_under_pmc"] = round(sum(dur.values()) / len(dur), 2)
for key, o in out.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in o and "SQ_BUSY_CU_CYCLES" in o:
        # MFMA busy cycles are summed over the SIMDs, SQ_BUSY_CU_CYCLES over the CUs that hold a wave of the launch:
        # the share of the occupied CUs' SIMD-cycles in which the matrix pipe is busy (16 cycles per 16x16x32 MFMA)
        o["mfma_util"] = round(o["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * o["SQ_BUSY_CU_CYCLES"]), 4)
    if "SQ_WAVE_CYCLES" in o:
        for k2 in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k2 in o:
                o[k2 + "_share_of_wave_cycles"] = round(o[k2] / o["SQ_WAVE_CYCLES"], 4)
    if "SQ_LDS_BANK_CONFLICT" in o and "SQ_LDS_IDX_ACTIVE" in o and o["SQ_LDS_IDX_ACTIVE"]:
        o["lds_conflict_share"] = round(o["SQ_LDS_BANK_CONFLICT"] / o["SQ_LDS_IDX_ACTIVE"], 4)
json.dump({"command": "rocprofv3 --kernel-trace --pmc <group> -- python bench.py $ARGS (one pass per group)", "per_launch": out},
          open("$R/gpurun_out/pmc_round_kernels.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf /tmp/prk_*
