cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > gpurun_out/r2h/ops_v2.log 2>&1; tail -3 gpurun_out/r2h/ops_v2.log
for k in ws v2; do
  LS_ATTN_KERNEL=$k timeout 300 python tools/bench_attn.py --L 16384 131072 --iters 50 2>/dev/null | tee gpurun_out/r2h/bench_attn_$k.log
done
