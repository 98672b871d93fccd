cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L=${L:-131072}
run() { name=$1; shift
  rm -rf /tmp/pa_$name; timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pa_$name -- python $R/tools/bench_attn.py --iters 3 --L $L > /tmp/pa_$name.log 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/pa_$name -name "*.db" | head -1) 2>&1 | grep -A12 "attn_partial" | grep -v "^==" 
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU
run c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC
run d SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM
run e SQ_IFETCH SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_WAVES
