# round 5, pass c: changed GPU tests; the warp-specialised kernel's ablations INSIDE the round (bench.py's event brackets around
# stage 1: the in-round clock); write-through partial stores A/B at 16k and 128k
set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5c
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -s -k "long or soak or saturating_key_behind" > $O/pytest_new.log 2>&1
grep -a "count/num\|rounds reproduce\|soak:\|passed\|failed\|one split" $O/pytest_new.log | tail -40
L=$PWD/longspec_amd/_lib
B="--steps 10 --warmup 3 --no-vanilla --no-cpu-baseline --no-cpu-round"
for v in default abl1 abl2 abl4 abl6 abl16 abl14 abl30 abl25 default; do
  if [ $v = default ]; then unset LONGSPEC_HIP_LIB; else export LONGSPEC_HIP_LIB=$L/liblongspec_hip_$v.so; fi
  echo "== $v" >> $O/ablate_inround.log
  timeout 200 python bench.py $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'ms_per_step': d['ms_per_step'], 'stage1_us': r['avg_launch_us'], 'attention_ms_per_round': d.get('attention_ms_per_round')}))" >> $O/ablate_inround.log 2>&1
done
unset LONGSPEC_HIP_LIB
for rep in 1 2; do
for v in default partwt; do
  if [ $v = default ]; then unset LONGSPEC_HIP_LIB; else export LONGSPEC_HIP_LIB=$L/liblongspec_hip_$v.so; fi
  for cfg in "--config 1" ""; do
    echo "== $v $cfg" >> $O/partwt.log
    timeout 300 python bench.py $cfg --steps 20 --warmup 5 --no-vanilla --no-cpu-baseline --no-cpu-round 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'ms_per_step': d['ms_per_step'], 'stage1_us': r['avg_launch_us'], 'attention_ms_per_round': d.get('attention_ms_per_round')}))" >> $O/partwt.log 2>&1
  done
done
done
unset LONGSPEC_HIP_LIB
cat $O/ablate_inround.log $O/partwt.log
