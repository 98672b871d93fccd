cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/longspec_amd/_lib/liblongspec_hip.so /tmp/lib_base.so
for v in base; do
  if [ $v != base ]; then cp $R/exp_libs/lib_$v.so $R/longspec_amd/_lib/liblongspec_hip.so; else cp /tmp/lib_base.so $R/longspec_amd/_lib/liblongspec_hip.so; fi
  for cfg in "16 14336 4096 --silu" "1 14336 4096 --silu" "32 14336 4096 --silu" "16 128256 4096" "1 128256 4096" "32 128256 4096" "74 128256 4096"; do
    set -- $cfg
    tag=${v}_M$1_N$2_K$3$4
    timeout 120 rocprofv3 --kernel-trace -d $R/gpurun_out/ex_$tag -- python $R/tools/prof_gemm.py --M $1 --N $2 --K $3 $4 --calls 8 > $R/gpurun_out/ex_$tag.log 2>&1
    db=$(find $R/gpurun_out/ex_$tag -name "*.db" | head -1)
    echo "== $tag: $(python $R/tools/rocprof_summary.py $db 2 | grep -i 'skinny\|Cijk' | awk -F'",' '{print substr($1,2,12), $2}' | tr '\n' ' ')"
  done
done
cp /tmp/lib_base.so $R/longspec_amd/_lib/liblongspec_hip.so
