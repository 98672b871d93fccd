# round 5, pass k: the RMSNorm in the projections' epilogue: operator test, end-to-end goldens, A/B inside the round
set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5k
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_linear.py -m gpu -x -q -k "norm_in_the_projection" > $O/pytest_norm.log 2>&1; tail -5 $O/pytest_norm.log
timeout 1500 python -m pytest tests/test_gpu_generate.py tests/test_gpu_linear.py -m gpu -q > $O/pytest_gen.log 2>&1; tail -5 $O/pytest_gen.log
run() {
  echo "== $*" >> $O/ab.log
  timeout 300 python bench.py $* --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-round 2>>$O/ab.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'value': d['value'], 'ms_per_step': d['ms_per_step'], 'tau': d['tau'], 'vanilla': d.get('vanilla_tokens_per_s'), 'stage1_us': r['avg_launch_us'], 'gemm_ms': d['roofline_gemm']['gemm_ms_per_round']}))" >> $O/ab.log 2>&1
}
for rep in 1 2 3; do
  LS_NORM_IN_EPILOGUE=1 run
  LS_NORM_IN_EPILOGUE=0 run
  LS_NORM_IN_EPILOGUE=1 run --config 1
  LS_NORM_IN_EPILOGUE=0 run --config 1
done
cat $O/ab.log
