# round 5, pass d: LS_WS_PF (operand fragments requested in front of the barrier) -- correctness of the variant library, then A/B
# inside the round at 128k and 16k; write-through partial stores A/B (pass c lost its default arm)
set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5e
mkdir -p $O
L=$PWD/longspec_amd/_lib
LONGSPEC_HIP_LIB=$L/liblongspec_hip_pf6.so timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "verify or full_size or saturating or sharded or lse or dominant or prefix" > $O/pytest_pf.log 2>&1
tail -4 $O/pytest_pf.log
run() {  # $1 = variant, rest = bench args
  v=$1; shift
  if [ $v = default ]; then unset LONGSPEC_HIP_LIB; else export LONGSPEC_HIP_LIB=$L/liblongspec_hip_$v.so; fi
  echo "== $v $*" >> $O/ab.log
  timeout 300 python bench.py $* --steps 20 --warmup 5 --no-vanilla --no-cpu-baseline --no-cpu-round 2>>$O/ab.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'ms_per_step': d['ms_per_step'], 'stage1_us': r['avg_launch_us'], 'frac': r['frac'], 'attention_ms_per_round': d.get('attention_ms_per_round')}))" >> $O/ab.log 2>&1
  unset LONGSPEC_HIP_LIB
}
for rep in 1 2; do
  for v in default pf6; do
    run $v
    run $v --config 1
  done
done
cat $O/ab.log
