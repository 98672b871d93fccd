# kernel-trace durations of ours vs hipBLASLt for the Llama-3-8B decode shapes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "74 14336 4096" "74 4096 14336" "74 4096 4096" "74 128256 4096" "32 14336 4096" ; do
  set -- $cfg
  tag=M$1_N$2_K$3
  timeout 120 rocprofv3 --kernel-trace -d $R/gpurun_out/tg_$tag -- python $R/tools/prof_gemm.py --M $1 --N $2 --K $3 --calls 8 $EXTRA > $R/gpurun_out/tg_$tag.log 2>&1
  db=$(find $R/gpurun_out/tg_$tag -name "*.db" | head -1)
  echo "== $tag"
  python $R/tools/rocprof_summary.py $db 2 | grep -i "skinny\|Cijk" | cut -c1-60,150-
done
