set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5p
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_topk.py tests/test_gpu_tree.py -m gpu -q > $O/pytest_topk.log 2>&1; tail -3 $O/pytest_topk.log
timeout 300 python tools/prof_topk.py > $O/prof_topk.log 2>&1; tail -20 $O/prof_topk.log
timeout 900 python -m pytest tests/test_gpu_generate.py -m gpu -q -k "not long and not soak" > $O/pytest_gen.log 2>&1; tail -3 $O/pytest_gen.log
for r in 1 2 3; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-round --no-vanilla | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
