set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5u
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -x -q > $O/ops.log 2>&1; tail -2 $O/ops.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-round | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['speedup_vs_vanilla'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
