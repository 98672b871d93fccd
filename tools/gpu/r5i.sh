# round 5, pass i: write-through partials through the compiler's buffer-store builtin: correctness (whole operator file), then A/B
set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5i
mkdir -p $O
L=$PWD/longspec_amd/_lib
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -q > $O/ops_default.log 2>&1; tail -3 $O/ops_default.log
run() {
  v=$1; shift
  if [ $v = default ]; then unset LONGSPEC_HIP_LIB; else export LONGSPEC_HIP_LIB=$L/liblongspec_hip_$v.so; fi
  echo "== $v $*" >> $O/ab.log
  timeout 300 python bench.py $* --steps 20 --warmup 5 --no-vanilla --no-cpu-baseline --no-cpu-round 2>>$O/ab.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'ms_per_step': d['ms_per_step'], 'tau': d['tau'], 'stage1_us': r['avg_launch_us'], 'frac': r['frac'], 'attention_ms_per_round': d.get('attention_ms_per_round')}))" >> $O/ab.log 2>&1
  unset LONGSPEC_HIP_LIB
}
for rep in 1 2 3; do
  for v in default nowt; do
    run $v
    run $v --config 1
  done
done
cat $O/ab.log
