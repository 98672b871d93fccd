# round 6, pass f: LS_WS_OPF=3 (O waves keep V^T in registers across the barrier, P just in time) -- parity, then A/B in the round
set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r6f
mkdir -p $O
V=$PWD/longspec_amd/_lib/liblongspec_hip_opf3.so
LONGSPEC_HIP_LIB=$V timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -x -q > $O/ops_opf3.log 2>&1; tail -3 $O/ops_opf3.log
run() {
  LONGSPEC_HIP_LIB=$2 timeout 600 python bench.py $3 --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-round --no-vanilla > $O/$1.json 2> $O/$1.err
  tail -1 $O/$1.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])" || tail -3 $O/$1.err
}
for i in 1 2 3; do
  run base_$i "" ""
  run opf3_$i $V ""
  run base_cfg1_$i "" "--config 1"
  run opf3_cfg1_$i $V "--config 1"
done
