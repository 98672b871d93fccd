# kernel-trace durations of ls_linear_fwd for "M N K [flags]" configs given on stdin
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
while read cfg; do
  set -- $cfg
  rm -rf /tmp/tg
  timeout 120 rocprofv3 --kernel-trace -d /tmp/tg -- python $R/tools/prof_gemm.py --M $1 --N $2 --K $3 $4 $5 $6 --calls 8 --no-torch > /tmp/tg.log 2>&1
  echo "$cfg: $(python $R/tools/rocprof_summary.py $(find /tmp/tg -name "*.db" | head -1) 2 | grep -i skinny | awk -F'",' '{print $2}')"
done
