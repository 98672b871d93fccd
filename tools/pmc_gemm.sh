cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name counters...
  name=$1; shift
  timeout 120 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pg_$name -- python $R/tools/prof_gemm.py --M 74 --N 14336 --K 4096 > $R/gpurun_out/pg_$name.log 2>&1
  db=$(find $R/gpurun_out/pg_$name -name "*.db" | head -1)
  python $R/tools/pmc_summary.py $db 2>&1 | grep -v "elementwise\|fill\|copy" 
}
run a SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
run b TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES
run c TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_REQ
run d TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_TCC_READ_REQ_LATENCY TCP_TCP_LATENCY
run e MemUnitStalled TCP_TA_TCP_STATE_READ TCC_TAG_STALL TCC_EA0_RDREQ_DRAM_CREDIT_STALL SQ_INSTS_VMEM_RD
