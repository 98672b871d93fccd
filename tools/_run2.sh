set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > gpurun_out/r2b/ops_v2.log 2>&1; echo "ops rc=$?" >> gpurun_out/r2b/ops_v2.log
tail -5 gpurun_out/r2b/ops_v2.log
for k in ws v2; do
  LS_ATTN_KERNEL=$k timeout 300 python tools/bench_attn.py --L 16384 131072 --iters 50 > gpurun_out/r2b/bench_attn_$k.log 2>&1
  cat gpurun_out/r2b/bench_attn_$k.log
done
