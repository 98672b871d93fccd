set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r6b
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_head.py -m gpu -x -q > $O/head.log 2>&1; tail -3 $O/head.log
timeout 1500 python -m pytest tests/test_gpu_generate.py tests/test_gpu_dist.py -m gpu -x -q > $O/gen.log 2>&1; tail -3 $O/gen.log
for i in 1 2; do
for ph in 0 1; do
LONGSPEC_PASS_HEAD=$ph timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-round > $O/bench_ph${ph}_$i.json 2> $O/bench_ph${ph}_$i.err; tail -1 $O/bench_ph${ph}_$i.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('ph$ph', d['value'], d['ms_per_step'], d['vanilla_tokens_per_s'], d['speedup_vs_vanilla'], d['roofline']['avg_launch_us'], d['roofline_gemm']['gemm_ms_per_round'])"
done
done
