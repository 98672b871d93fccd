# round 5, pass t: LS_WS_PF with the DMA issue at the END of the step (fragments prefetched, not waited for; DMA behind the wave's
# matrix work): correctness, A/B inside the round, phase stamps
set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5t
mkdir -p $O
L=$PWD/longspec_amd/_lib
LONGSPEC_HIP_LIB=$L/liblongspec_hip_pf7.so timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "verify or full_size or saturating or sharded or lse or dominant or prefix or sink" > $O/pytest_pf7.log 2>&1
tail -3 $O/pytest_pf7.log
run() {
  v=$1; shift
  if [ $v = default ]; then unset LONGSPEC_HIP_LIB; else export LONGSPEC_HIP_LIB=$L/liblongspec_hip_$v.so; fi
  echo "== $v $*" >> $O/ab.log
  timeout 300 python bench.py $* --steps 20 --warmup 5 --no-vanilla --no-cpu-baseline --no-cpu-round 2>>$O/ab.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'ms_per_step': d['ms_per_step'], 'tau': d['tau'], 'stage1_us': r['avg_launch_us'], 'frac': r['frac']}))" >> $O/ab.log 2>&1
  unset LONGSPEC_HIP_LIB
}
for rep in 1 2 3; do
  for v in default pf7; do
    run $v
    run $v --config 1
  done
done
for v in wsprof pf7prof; do
  for LL in 131072 16384; do
    echo "== $v L=$LL" >> $O/wsprof.log
    L=$LL LONGSPEC_HIP_LIB=$PWD/longspec_amd/_lib/liblongspec_hip_$v.so timeout 200 python tools/ws_prof.py >> $O/wsprof.log 2>>$O/wsprof.err
  done
done
cat $O/ab.log; cut -c1-700 $O/wsprof.log
