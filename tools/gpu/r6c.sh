set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r6c
mkdir -p $O
run() {  # name, PF_SET, PF_FRAC
  PF_SET=$2 PF_FRAC=$3 timeout 600 python tools/ab_mall_prefetch.py --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-round --no-vanilla > $O/$1.json 2> $O/$1.err
  tail -1 $O/$1.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline_gemm']['gemm_ms_per_round'])" || tail -5 $O/$1.err
}
run none_1 none 1.0
run o_1 o 1.0
run o_qkv_1 o,qkv 1.0
run o_gu_1 o,gu 1.0
run all_1 o,gu,d,qkv 1.0
run all_half o,gu,d,qkv 0.5
run none_2 none 1.0
run gu_d_1 gu,d 1.0
