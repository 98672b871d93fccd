# round 5, pass g: LS_WS_SDMA (only the S waves request the K/V stream): correctness, A/B inside the round, phase stamps
set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5g
mkdir -p $O
L=$PWD/longspec_amd/_lib
LONGSPEC_HIP_LIB=$L/liblongspec_hip_sdma.so timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "verify or full_size or saturating or sharded or lse or dominant or prefix" > $O/pytest_sdma.log 2>&1
tail -3 $O/pytest_sdma.log
run() {
  v=$1; shift
  if [ $v = default ]; then unset LONGSPEC_HIP_LIB; else export LONGSPEC_HIP_LIB=$L/liblongspec_hip_$v.so; fi
  echo "== $v $*" >> $O/ab.log
  timeout 300 python bench.py $* --steps 20 --warmup 5 --no-vanilla --no-cpu-baseline --no-cpu-round 2>>$O/ab.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'ms_per_step': d['ms_per_step'], 'stage1_us': r['avg_launch_us'], 'frac': r['frac'], 'attention_ms_per_round': d.get('attention_ms_per_round')}))" >> $O/ab.log 2>&1
  unset LONGSPEC_HIP_LIB
}
for rep in 1 2 3; do
  for v in default sdma; do
    run $v
    run $v --config 1
  done
done
for v in wsprof sdmaprof; do
  for LL in 131072 16384; do
    echo "== $v L=$LL" >> $O/wsprof.log
    L=$LL LONGSPEC_HIP_LIB=$PWD/longspec_amd/_lib/liblongspec_hip_$v.so timeout 200 python tools/ws_prof.py >> $O/wsprof.log 2>>$O/wsprof.err
  done
done
cat $O/ab.log $O/wsprof.log
